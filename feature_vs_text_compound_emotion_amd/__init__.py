"""MI355X-native hot path for feature-based compound emotion recognition.

Host-side mirror of the reference's ``models/`` surface (LFAN & co.) on top of
libcer_hip.so -- hand-written HIP kernels for gfx950 behind the C-ABI declared
in ``include/cer_hip.h``.
"""
__version__ = "0.1.0"
