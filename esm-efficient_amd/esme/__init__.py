"""MI355X-native drop-in for the packed forward path of uci-cbcl/esm-efficient.

    from esme import ESM, ESM2, ESMC, tokenize

(the same names the reference exports from `esme/__init__.py:1-4`; ESM-1b/1v are
outside the hot-path scope).  Put `esm-efficient_amd/` on sys.path.
"""
from esme.alphabet import tokenize, tokenize_unpad          # noqa: F401
from esme.esm import ESM, ESM2, ESMC                        # noqa: F401

__all__ = ['ESM', 'ESM2', 'ESMC', 'tokenize', 'tokenize_unpad']
__version__ = '0.1.0'
