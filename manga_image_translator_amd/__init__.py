"""MI355X-native dense-stage engine for manga-image-translator (detect -> OCR -> inpaint).

Hand-written gfx950 HIP kernels behind a C-ABI (``include/mit_hip.h``), driven from Python
host code that mirrors the reference's Detector / Ocr / Inpainter plugin interfaces.
"""
__version__ = "0.1.0"
