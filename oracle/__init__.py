"""CPU oracle for the DKT hot path -- test infrastructure only (see dkt_oracle.py header).
PARITY UNPINNED: no reference-run outputs or reference tests exist for this path."""
