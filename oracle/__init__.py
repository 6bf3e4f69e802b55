"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's algorithm for the hot path (ref_ops.c / ref_ops.py:
the 18 native ops + segment_cumsum; model_oracle.py: the torch half of FourierGridModel / DirectVoxGO) and the stub
installer that lets the reference's own Python run on a CPU.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by unboundednerfpytorch_amd."""
