"""NativeModuleSwap: how a native implementation takes the place of a CHILD ``nn.Module`` of the reference model.

``model.encoder`` / ``model.detokenizer`` are registered sub-modules of ``AceStepConditionGenerationModel``
(modeling_acestep_v15_base.py:1571-1573).  ``nn.Module.__setattr__`` refuses to assign a non-Module to such a name, and the
call sites (``prepare_condition``, base.py:1626-1633, 1646-1647; handler/audio_codes.py:49-66) sit OUTSIDE the diffusion seam's
try block (handler/service_generate_execute.py:123), so nothing would catch a native failure there.  This wrapper is an
``nn.Module`` (assignable), keeps the module it replaces as a registered sub-module (state_dict keys unchanged), and delegates to it whenever the native call raises - the same
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
        object.__setattr__(self, "_native", native_impl)
        # The kept module is a REAL sub-module (``parent.to()`` / ``half()`` / ``parameters()`` / ``load_state_dict()`` all recurse
        # through the registry: a fallback kept outside it would stay on the old device and miss reloaded weights - exactly when
        # it is needed).  Two hooks keep the parent's state_dict keys the reference's own (no ``_reference.`` segment).
        self._reference = reference_module
        self._swap_name = name
        self.native_calls = 0
        self.native_failures = 0
        self._register_state_dict_hook(self._strip_reference_prefix)
        self._register_load_state_dict_pre_hook(self._add_reference_prefix)
        self.register_load_state_dict_post_hook(self._reload_native)

    @staticmethod
    def _strip_reference_prefix(module, state_dict, prefix, local_metadata):
        inner = prefix + "_reference."
        for k in [k for k in state_dict if k.startswith(inner)]:
            state_dict[prefix + k[len(inner):]] = state_dict.pop(k)
        return state_dict

    @staticmethod
    def _add_reference_prefix(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        inner = prefix + "_reference."
        for k in [k for k in state_dict if k.startswith(prefix) and not k.startswith(inner)]:
            state_dict[inner + k[len(prefix):]] = state_dict.pop(k)

    @staticmethod
    def _reload_native(module, incompatible_keys):
        # weights changed under the wrapper (parent.load_state_dict): the native copy is re-packed from the kept module
        native = object.__getattribute__(module, "_native")
        if hasattr(native, "load_state_dict"):
            try:
                native.load_state_dict(module._reference.state_dict())
            except Exception as exc:  # keep serving through the PyTorch module
                logger.warning("[native-%s] re-packing after load_state_dict failed (%s: %s)", module._swap_name, type(exc).__name__, exc)

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
            if item == "_reference":
                raise
            return getattr(super().__getattr__("_reference"), item)


def swap_in(parent: torch.nn.Module, attr: str, native_cls, device, out_dtype: torch.dtype = torch.float32) -> NativeModuleSwap:
    """``parent.<attr> = NativeModuleSwap(native_cls.from_reference(parent.<attr>), parent.<attr>)``; returns the wrapper.
    Raises whatever ``from_reference`` raises (caller logs and keeps the PyTorch module, like ``_init_native_dit``)."""
    ref = getattr(parent, attr)
    if isinstance(ref, NativeModuleSwap):  # re-entry (initialize_service may be called again)
        ref = ref._reference
    wrapper = NativeModuleSwap(native_cls.from_reference(ref, device, out_dtype), ref, attr)
    setattr(parent, attr, wrapper)
    return wrapper
