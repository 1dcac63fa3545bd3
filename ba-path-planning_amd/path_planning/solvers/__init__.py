from .scp import SCP

__all__ = ["SCP"]
