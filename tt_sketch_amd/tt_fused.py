"""Fused fast path for TT / sum-of-TT inputs with TT DRMs (the north-star configuration).
Placeholder: returns None so that general_sketch composes the path from ttsk_gemm."""


def try_stream_sketch(tensor, left_drm, right_drm, method):
    return None
