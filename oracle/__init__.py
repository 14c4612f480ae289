"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the VINCE encoder + contrastive hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / timed baseline.
The product path (``vince_amd``) never imports this package and has no CPU fallback.

Pinning status: the reference (danielgordon10/vince) ships no tests, golden vectors or fixtures
for this path.  The oracle is therefore pinned against outputs of the
reference itself, imported on CPU in the build container by ``oracle/make_golden.py`` (which
needs ``/root/reference`` and the ``dg_util`` stand-ins in ``oracle/ref_harness.py``); the
resulting vectors are committed under ``tests/golden/`` and ``tests/test_oracle_golden.py``
checks this restatement against every one of them.
"""
