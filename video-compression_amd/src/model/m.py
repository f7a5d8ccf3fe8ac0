"""``src.model.m`` -- same import path as ICIP2024/src/model/m.py; ``FlowGuidedB`` runs on MI355X through libvc_hip.so."""
from vcamd.icip2024 import FlowGuidedB  # noqa: F401
