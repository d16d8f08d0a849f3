"""qinco_amd -- MI355X-native (gfx950) QINCo / QINCo2 encode-decode engine.

Python host over libqinco_hip.so (hand-written HIP kernels behind a C ABI, include/qinco_hip.h).
"""
from .config import BASELINE_CONFIGS, QincoConfig, preset  # noqa: F401
from .synth import apply_regime, regime_vectors, synth_codes, synth_state_dict, synth_vectors  # noqa: F401

__all__ = ["QincoConfig", "preset", "BASELINE_CONFIGS", "QincoEngine", "QINCoHIP", "synth_state_dict",
           "synth_vectors", "synth_codes", "apply_regime", "regime_vectors"]


def __getattr__(name):  # engine / model import the HIP library lazily
    if name == "QincoEngine":
        from .engine import QincoEngine
        return QincoEngine
    if name == "QINCoHIP":
        from .model import QINCoHIP
        return QINCoHIP
    raise AttributeError(name)
