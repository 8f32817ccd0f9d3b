"""NativeModuleSwap: how a native implementation takes the place of a CHILD ``nn.Module`` of the reference model.

``model.encoder`` / ``model.detokenizer`` are registered sub-modules of ``AceStepConditionGenerationModel``
(modeling_acestep_v15_base.py:1571-1573).  ``nn.Module.__setattr__`` refuses to assign a non-Module to such a name, and the
call sites (``prepare_condition``, base.py:1626-1633, 1646-1647; handler/audio_codes.py:49-66) sit OUTSIDE the diffusion seam's
try block (handler/service_generate_execute.py:123), so nothing would catch a native failure there.  This wrapper is an
``nn.Module`` (assignable), keeps the module it replaces, and delegates to it whenever the native call raises - the same
"try native -> log -> PyTorch" policy as the DiT / VAE seams (handler/service_generate_execute.py:189-191, handler/vae_decode.py:44-48).
"""
from __future__ import annotations

import logging

import torch

logger = logging.getLogger("ace355")


class NativeModuleSwap(torch.nn.Module):
    """Callable like the reference module; ``native_impl`` first, the kept reference module on ValueError / RuntimeError /
    NotImplementedError (unsupported mask form, missing library, failed HIP call)."""

    def __init__(self, native_impl, reference_module: torch.nn.Module, name: str = "module"):
        super().__init__()
        # kept OUT of the module registry: parent.state_dict() keys, parameter counts and optimiser groups stay exactly the
        # reference's.  (The native path refuses CPU offload, so nobody moves the parent between devices behind our back.)
        object.__setattr__(self, "_native", native_impl)
        object.__setattr__(self, "_reference", reference_module)
        self._swap_name = name
        self.native_calls = 0
        self.native_failures = 0

    def forward(self, *args, **kwargs):
        try:
            out = self._native(*args, **kwargs)
            self.native_calls += 1
            return out
        except (ValueError, RuntimeError, NotImplementedError) as exc:
            self.native_failures += 1
            if self.native_failures <= 3:
                logger.warning("[native-%s] %s: %s; running the PyTorch module for this call", self._swap_name, type(exc).__name__, exc)
            return self._reference(*args, **kwargs)

    def __getattr__(self, item):
        # attributes of the module it stands in for (config, layers, dtype helpers ...)
        try:
            return super().__getattr__(item)
        except AttributeError:
            return getattr(object.__getattribute__(self, "_reference"), item)

    def state_dict(self, *args, **kwargs):
        return self._reference.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        out = self._reference.load_state_dict(state_dict, *args, **kwargs)
        if hasattr(self._native, "load_state_dict"):
            self._native.load_state_dict(self._reference.state_dict())
        return out


def swap_in(parent: torch.nn.Module, attr: str, native_cls, device, out_dtype: torch.dtype = torch.float32) -> NativeModuleSwap:
    """``parent.<attr> = NativeModuleSwap(native_cls.from_reference(parent.<attr>), parent.<attr>)``; returns the wrapper.
    Raises whatever ``from_reference`` raises (caller logs and keeps the PyTorch module, like ``_init_native_dit``)."""
    ref = getattr(parent, attr)
    if isinstance(ref, NativeModuleSwap):  # re-entry (initialize_service may be called again)
        ref = ref._reference
    wrapper = NativeModuleSwap(native_cls.from_reference(ref, device, out_dtype), ref, attr)
    setattr(parent, attr, wrapper)
    return wrapper
