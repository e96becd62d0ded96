"""CPU oracle for the encode + kNN hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  Nothing under
``cellcomm_amd/`` imports it; the product path has no CPU fallback.

Parity status:
  * loader / sampling / schema helpers — pinned by the reference's own known
    answers (tests/golden/reference_known_answers.json).
  * encoder forward — restates Keras 2.4 op definitions called at
    reference src/bigan_cont.py:28-41 and src/bigan_classify.py:28-40;
    tensorflow==2.4.0 (reference requirements.txt:3) is not installable here and
    no reference test pins an encoder output value: NUMERIC PARITY UNPINNED.
  * kNN — the reference has no kNN code at all: PARITY UNPINNED;
    cross-checked against sklearn.neighbors brute force on tie-free data.
"""
