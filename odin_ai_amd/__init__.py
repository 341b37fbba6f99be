"""odin_ai_amd -- MI355X-native VAE training step behind the odin-ai VariationalAutoencoder API."""
__version__ = '0.1.0'
