"""CPU oracle for the PanSt3R inference forward path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain-PyTorch fp32 *restatement* of the reference algorithm
(naver/panst3r v0.2.1, plus the un-vendored must3r / croco / dust3r pieces it
calls).  It exists only to check the HIP path in ``panst3r_amd``.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``panst3r_amd/`` imports
it, and the product path never falls back to it.

Parity status
-------------
* Panoptic half (InputMixer glue, PixelShuffle / LoftUp upscalers,
  MaskTransformer incl. the query x pixel einsum, PositionEmbeddingSine,
  PanopticDecoder, DinoV2Encoder wrapper): **pinned** against golden vectors
  generated in the build container by importing the reference's own modules
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* CroCo ViT encoder, MUSt3R decoder + memory bank, RoPE2D, croco blocks:
  the upstream packages are absent from /root/reference and un-pinned
  (pyproject.toml:14) => **parity unpinned**.  They are restated from the
  call-site contracts (engine/must3r.py:28-129, panst3r.py:65-86,205-234,
  configs/base.yaml:6-15, model/blocks.py:9-35) and the CroCo / DUSt3R /
  MUSt3R papers, and checked with known-answer tests only.
"""
