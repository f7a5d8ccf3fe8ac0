"""Drop-in import names of the reference's LHBDC package: ``from model import m`` (LHBDC/encode_B.py:16)."""
