"""CPU oracle for the bucketMul hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``effort_amd``) never does; it fails loudly without its
HIP library instead of falling back to anything here.

* ``oracle.cpu``        -- ctypes binding of ``effort_oracle.c`` (FP16 + Q4 multiply, converter,
                           cutoff, dispatch, dense GEMV, cosine).  FP16 parity vs the Swift/Metal
                           reference is UNPINNED (nothing of it runs here); see the C header.
* ``oracle.q4_layout``  -- numpy restatement of ``q4_draft.convert`` (pinned by tests/golden/).
"""
