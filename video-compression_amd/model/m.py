"""``model.m`` -- same import path as LHBDC/model/m.py; ``Model`` runs on MI355X through libvc_hip.so."""
from vcamd.lhbdc import Model  # noqa: F401
