"""CPU oracle for the GATOR forward path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm (kasvii/GATOR,
``lib/models/{GATOR,GAT,MDR}.py`` and helpers).  It exists to *check* the HIP
path; it is never the thing shipped or measured as the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``gator_amd/`` imports it, and the product
path raises when the HIP library is missing instead of falling back to this.

Parity pinning: the reference repo holds no tests or golden vectors for this
path.  The oracle is therefore pinned against outputs of
the reference itself, produced in the dev container by ``tools/gen_golden.py``
(shimmed import of /root/reference) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks the oracle against them.
"""
