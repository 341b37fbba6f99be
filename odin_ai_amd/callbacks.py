"""Training callbacks of the reference's VAE examples (examples/vae/utils.py:300-467) on the HIP path.

Only the callback SURVEY.md 8(f2) names is built: `Callback.save_best_llk` -- the mean reconstruction
log-likelihood log p(x | z), z ~ q(z|x), over a validation set, logged as `valid/llk`, and a checkpoint
(`model.save_weights(overwrite=True)`) whenever it improves.  The plotting callbacks of that file
(latent traversals, pixel likelihood maps) are matplotlib / TensorBoard-image code outside the hot path.
"""
from typing import Callable, Dict, Iterable, Optional, Union

import numpy as np
import torch


class Callback:
  """examples/vae/utils.py:300 `class Callback`: static methods handed to `fit(on_valid_end=...)`."""

  n_valid_batches: int = 200          # examples/vae/utils.py: `_n_valid_batches`
  _best: Dict[int, float] = {}        # examples/vae/utils.py: `_best = defaultdict(lambda: -np.inf)`

  @staticmethod
  def _batches(valid_ds, batch_size: int, device) -> Iterable[torch.Tensor]:
    if torch.is_tensor(valid_ds) or isinstance(valid_ds, np.ndarray):
      data = torch.as_tensor(valid_ds, dtype=torch.float32).to(device)
      bs = min(int(batch_size), data.shape[0])
      for i in range(0, data.shape[0] - bs + 1, bs):
        yield data[i:i + bs].contiguous()
    else:
      for b in valid_ds:
        if isinstance(b, (tuple, list)):  # (x, y) pairs of a labelled dataset: the VAE sees x
          b = b[0]
        yield torch.as_tensor(b, dtype=torch.float32).to(device)

  @staticmethod
  def save_best_llk(model, valid_ds, batch_size: int = 64, n_valid_batches: Optional[int] = None,
                    log: Optional[Callable[[str], None]] = print) -> float:
    """examples/vae/utils.py:446-467.  For up to `n_valid_batches` batches: (px, qz) = model(x),
    llk = px.log_prob(x) (one posterior sample, event shape = image shape -> [B]); the mean over all samples is
    written as the TensorBoard scalar `valid/llk` (when the model logs to a `logdir`) and compared with the best
    value seen for THIS model object; an improvement saves the weights to `model.path`.  Returns the llk."""
    n_max = Callback.n_valid_batches if n_valid_batches is None else int(n_valid_batches)
    vals = []
    for i, x in enumerate(Callback._batches(valid_ds, batch_size, model.device)):
      if i >= n_max:
        break
      px, _ = model(x)
      vals.append(px.log_prob(x).reshape(-1).double())
    if not vals:
      raise ValueError('save_best_llk: the validation set yielded no batch')
    llk = float(torch.cat(vals).mean().item())
    ev = getattr(model, '_events', None)
    if ev is not None:
      ev.scalar('valid/llk', llk, model.step)
      ev.flush()
    key = id(model)
    best = Callback._best.get(key, -np.inf)
    if llk > best:
      Callback._best[key] = llk
      model.save_weights(overwrite=True)
      if log is not None:
        log(f'best llk: {llk:.2f}')
    elif log is not None:
      log(f'worse llk: {llk:.2f} vs best: {best:.2f}')
    return llk
