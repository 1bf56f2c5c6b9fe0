"""Drop-in mirrors of the reference's `pytorch_model` sub-packages for the matching hot path.

Same class names, constructor arguments, defaults, ValueErrors, buffer names and forward()
signatures as reference pytorch_model/{detector,utils,descriptor,matching,feature_detection};
the arithmetic runs in the hand-written gfx950 kernels behind include/mi355x_match.h.
Inputs must be GPU tensors (there is no CPU fallback).
"""
