"""MI355X-native batched playroom simulator behind the reference's gym surface (hot path only)."""
from .vec_env import VecPlayEnv  # noqa: F401

__all__ = ['VecPlayEnv']
