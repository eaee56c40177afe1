"""MI355X-native drop-in for the packed forward path of uci-cbcl/esm-efficient.

    from esme import ESM, ESM2, ESMC, tokenize

(the names the reference exports from `esme/__init__.py:1-4`, plus the ESM-1b/1v classes).  Put `esm-efficient_amd/` on sys.path.
"""
from esme.alphabet import tokenize, tokenize_unpad          # noqa: F401
from esme.esm import ESM, ESM2, ESM1b, ESM1v, ESMC          # noqa: F401

__all__ = ['ESM', 'ESM2', 'ESM1b', 'ESM1v', 'ESMC', 'tokenize', 'tokenize_unpad']
__version__ = '0.1.0'
