"""CPU oracle for the ECoG->text seq2seq hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (jgmakin/ecog2txt @ 2024-12-20) delegates this
path to the un-vendored, un-pinned TF1.x package `machine_learning`
(ecog2txt/trainers.py:32-33, 126-135) and ships no tests or golden vectors
(SURVEY.md section 4, 8c).  This oracle is a NumPy fp64 restatement of the
normative spec in DESIGN.md, cross-checked against an independent torch-CPU
autograd model in tests/test_oracle_vs_torch.py.  cpu_step.cpp / cpu_step.py: the C++17 / OpenMP fp32
restatement of the train step that bench.py's cpu_baseline times, pinned against
the NumPy oracle by tests/test_cpu_step.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (ecog2txt_amd) must never import it.
"""
