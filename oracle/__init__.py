"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

CPU restatement of the reference's hot path (see keras_ops.py header for what pins it).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
