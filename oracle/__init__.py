"""CPU oracle for the GNNML1/GNNML3 spectral layer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / reported CPU baseline -- never as the
thing that is shipped or measured as the MI355X path.

The oracle is a restatement (our own code, op-for-op and in the same order) of
the reference algorithm:

  * ``spect_conv_oracle.py``   <- /root/reference/libs/spect_conv.py
       (SpectConv :23-103, SpectConCatConv :105-165, ML3Layer :182-212) plus the
       slice of pytorch_geometric==1.6.1 ``MessagePassing.propagate`` those
       classes call (gather -> message -> scatter-add; un-vendored dependency,
       pinned only in prose at /root/reference/README.md:9-11).
  * ``spectral_design_oracle.py`` <- /root/reference/libs/utils.py:525-626
  * ``csr_oracle.py``            integer-exact COO -> CSR-by-destination.
  * ``models_oracle.py``         GNNML1/GNNML3 assemblies of the five BASELINE
       configs (mutag.py:214-309, counting.py:335-372, Zinc12k.py:248-345,
       sr25.py:205-278, libs/models_tf.py:191-268).

Pinning: the reference tree has no tests and no golden vectors,
so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, produced in the
build container by importing /root/reference/libs/{spect_conv,utils}.py
unmodified under a minimal ``torch_geometric`` stand-in (``oracle/make_golden.py``,
committed; vectors in ``tests/golden/*.npz``).  ``tests/test_oracle_golden.py``
checks the oracle against every one of those vectors.
"""
