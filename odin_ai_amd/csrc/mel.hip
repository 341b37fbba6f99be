// mel.hip -- speech front-end (placeholder until the batched FFT kernel lands)
#include "odin_device.h"
#include "odin_internal.h"

extern "C" int odin_stft_mel_db(const float*, const float*, const float*, float*, int, int, int,
                                int, int, int, float, float, int, void*) {
  return odin_fail(-3, "odin_stft_mel_db: not implemented yet");
}
