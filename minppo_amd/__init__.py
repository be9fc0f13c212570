"""MI355X-native PPO engine with the minppo train/env/config surface."""

__version__ = "0.1.0"
