"""The parts of bench.py (repo root) that are not its timed core: shared set-up and the encode run (common), the roofline objects
(roofline), the reference side of the parity checks and CPU baselines (reference), every other leg of the line (extras)."""
