"""``src`` -- same import path as ICIP2024/src (the reference runs ``from src.model import m``)."""
