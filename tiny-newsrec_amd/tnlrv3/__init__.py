"""Checkpoint-format helpers of the UniLMv2 (TuringNLRv3) encoder: the counterpart of the reference's tnlrv3/ package
as far as the training hot path needs it (weight import; the model itself runs in engine.py)."""
