"""CPU oracle for the ACE-Step 1.5 denoise + decode hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 restatement of the
reference's algorithm for SURVEY.md section 8(a) rows D1-D19 (DiT sampler) and V1-V7
(Oobleck VAE decoder + tiling + post-processing).  It exists so that the HIP path
can be checked against the reference's arithmetic on a box where the reference
itself cannot travel.

Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing in the product package may import it; the product
path fails loudly when the HIP library is missing instead of falling back here.

Pinning status (see DESIGN.md "Oracle"):
* DiT / sampler / APG / ADG (``dit.py``, ``sampler.py``, ``apg.py``): PINNED against
  outputs of the imported reference (``/root/reference/acestep/models/base``),
  captured by ``tests/golden/make_golden.py`` into ``tests/golden/*.npz``.
* Tiling + post-processing (``tiling.py``): PINNED against the imported
  ``acestep/core/generation/handler/vae_decode_chunks.py``.
* Oobleck decoder arithmetic (``oobleck.py``): **parity unpinned** - the
  algorithm lives in third-party ``diffusers`` (unpinned in the reference's
  pyproject.toml:25), absent from /root/reference and from this image.  It is
  restated from the reference's in-tree MLX restatement
  (acestep/models/mlx/vae_model.py, vae_convert.py) and self-checked against
  ``torch.nn.utils.parametrizations.weight_norm`` + ``F.conv1d`` compositions.
"""
