"""MI355X-native hot path of MB-VAE-DTI: host-side mirror of the reference module API over libmbvae_hip.so."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
