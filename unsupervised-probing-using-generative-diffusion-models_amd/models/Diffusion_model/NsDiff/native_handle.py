"""Lazily created mpv_ctx for one model object: uploads the weights on first use and again
whenever the owner marks them dirty (load_state_dict, .to(), refresh_native())."""
from __future__ import annotations

import weakref

import torch

from .... import _native


class NativeHandle:
    def __init__(self):
        self._owner = None
        self.ctx = None
        self.dirty = True

    def bind(self, owner):
        self._owner = weakref.ref(owner)

    def mark_dirty(self):
        self.dirty = True

    def context(self, _caller=None) -> "_native.Context":
        owner = self._owner() if self._owner else None
        if owner is None:
            raise RuntimeError("native handle is not bound to a model")
        device = owner.scaler_mean.device
        if device.type != "cuda":
            raise _native.MpvError(
                "model is on {!r}: the NsDiff sampling path is implemented only as HIP kernels for gfx950; "
                "move the model to a ROCm device (there is no CPU fallback)".format(str(device)))
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.ctx is None or self.ctx.device != device:
            if self.ctx is not None:
                self.ctx.close()
            self.ctx = _native.Context(device)
            self.dirty = True
        if self.dirty:
            owner._upload_native(self.ctx)
            self.dirty = False
        return self.ctx
