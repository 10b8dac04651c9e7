"""ORACLE = test infrastructure.  CPU restatement of the reference algorithm for the hot path
(SURVEY.md 8(c)).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package; the product (segdino3d_amd) never does."""
