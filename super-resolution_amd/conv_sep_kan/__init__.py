"""Drop-in for the reference's `conv_sep_kan` package: `from conv_sep_kan import ConvSepKan`
(conv_sep_kan/__init__.py:1) resolves to the MI355X-native module."""
from kan_sr.module import ConvSepKan  # noqa: F401
