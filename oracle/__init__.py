"""CPU oracle for the PPO + TransformerXL hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the reported CPU
baseline -- never as the thing that is shipped or measured as the MI355X path.

The oracle is a from-scratch *functional* restatement (plain functions over a
``state_dict`` of fp32 CPU tensors) of the reference algorithm in
MarcoMeter/episodic-transformer-memory-ppo.  Every function cites the
reference file:line it restates.  It is pinned against golden vectors that
were produced by importing the real reference in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.
"""
