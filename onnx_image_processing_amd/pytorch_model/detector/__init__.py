from .shi_tomasi import ShiTomasiScore

__all__ = ["ShiTomasiScore"]
