"""CPU oracle for the ARP-DT hot paths.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package -- and there only as the checker, never as the thing measured or
shipped.  The product (``arp_amd``) never imports ``oracle``.

Parity status: the reference (csmile-1006/ARP) ships no tests, golden vectors or fixtures for
either hot path, and neither path can be imported in this image (jax, flax,
clip, h5py, torchvision are absent).  Parity is therefore **unpinned by the reference's own
tests**.  The restatements here are pinned instead against independent implementations that *are*
available: PIL 12.2 (bicubic resize, bit-exact), HuggingFace ``transformers.CLIPModel`` with
``hidden_act="quick_gelu"`` (an independent implementation of openai/CLIP, the third-party
dependency that holds the arithmetic of path 1, pinned by the reference at commit d50d76d), and a
torch-autograd mirror plus fp64 finite differences for the policy path.  See DESIGN.md section 3.
"""
