"""CPU oracle for the EzAudio denoising path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference's algorithm for the one
hot path this repo accelerates (MaskDiT/UDiT forward, CFG + rescale, DDIM
v-prediction step).  Every function cites the reference file:line it follows.

It is the *checker*, never the product:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``ezaudio_amd/`` imports it, and the product path raises if
    the HIP extension is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle.dit`` / ``oracle.sampler``: pinned against the reference's own
    modules, imported from /root/reference in the build container by
    ``oracle/mint_golden.py`` -> ``tests/golden/*.npz``.
  * ``oracle.ddim``: the reference calls ``diffusers.DDIMScheduler`` which is
    not vendored, not pinned and not installable here, and the reference holds
    no tests for it -> PARITY UNPINNED at that boundary.  The restatement is
    checked against the anchors recorded in SURVEY.md section 8a row S.
"""
