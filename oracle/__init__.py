"""ORACLE: CPU restatement of the reference's algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; nothing under compare_gan_amd/ does.  The reference (TensorFlow 1.x)
cannot be imported or run in this environment, so this is a "port"-kind
oracle, pinned against the reference's own golden vectors where they exist
(tests/test_oracle_pins.py) and marked "parity unpinned" elsewhere.
"""
