"""oracle/ -- TEST INFRASTRUCTURE, not product code.

A CPU fp32 restatement (plain torch functional ops / numpy) of the reference's ProPainter
inference algorithm, used as the checker for the HIP path.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  Parity pin: every function here is checked
against outputs of the reference itself (imported on CPU with the shims of
tests/golden/ref_import.py) through the committed fixtures in tests/golden/ -- the reference
ships no tests or golden vectors of its own.
"""
