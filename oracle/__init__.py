"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch fp32) of the lucidrains/phenaki-pytorch hot path
(C-ViViT encode/decode + MaskGIT sampling), plus the tooling that pins it to
the real reference.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from this package, and only as the
CHECKER -- never as the thing measured or shipped.  The product package
(``phenaki_pytorch_amd``) never imports it and has no CPU fallback.

Parity pinning: the reference ships no tests / golden vectors,
so ``oracle/make_golden.py`` runs the *real* reference (imported read-only
from /root/reference through ``oracle/ref_shim.py``) on seeded inputs and
commits the outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those files.  The one piece of arithmetic
that is NOT in /root/reference -- ``vector_quantize_pytorch.LFQ`` (pinned only
as ``>=1.11.8`` in the reference's setup.py:33) -- is restated in
``oracle/lfq.py`` from the library's published behaviour; that sub-step is
"parity unpinned" (no upstream vectors exist for it).
"""
