"""CPU oracle for the depth-training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU / numpy fp32
restatement of the reference algorithm (zzzxxxttt/SimpleDepthEstimation) for the
path named in BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it -- as the checker / reported
baseline, never as the product.  The product package
(``simpledepthestimation_amd``) must never import from here.

Parity pinning: every function here is checked (tests/test_oracle_golden.py)
against golden vectors generated in the build container by running the
reference's own, unmodified Python files (oracle/gen_golden.py, which imports
/root/reference through stub parent packages).  Where the arithmetic lives in a
third-party dependency that is not vendored in the reference (torchvision 0.9
ResNet wiring, README.md:L26), this is stated at the function ("parity
unpinned" beyond torch's own CPU conv/batch_norm ops).
"""
