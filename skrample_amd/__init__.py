"""skrample_amd -- MI355X-native sampler-step engine behind skrample's sampler / scheduler API."""

__version__ = "0.1.0"
