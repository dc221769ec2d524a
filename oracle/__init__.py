"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the AnyLoc-VLAD-DINOv2 hot
path (DINOv2 facet extraction -> VLAD -> flat top-k retrieval -> k-means
vocabulary).  Nothing under ``anyloc_amd/`` or ``utilities.py`` (the product)
may import this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

Parity pinning (see DESIGN.md "Oracle"): the reference ships no tests, golden
vectors or fixtures for this path, so the pins are made
here: ``oracle/ref_loader.py`` executes the reference's own ``utilities.py``
verbatim (possible only where ``/root/reference`` exists) and
``oracle/make_golden.py`` records its outputs under ``tests/golden/``; the
restatements in this package are checked against those recordings by
``tests/test_oracle_golden.py``.  The three third-party pieces that are not
vendored in the reference (facebookresearch/dinov2 @ main via torch.hub,
fast-pytorch-kmeans==0.1.6, faiss==1.7.2) are restated from their published
algorithms; DINOv2 is additionally cross-checked against the independent
``transformers`` implementation (tests/test_oracle_dinov2_hf.py).

``oracle/c/`` holds the same VLAD / k-means / flat-search / recall arithmetic as scalar C with double accumulators
(``anyloc_oracle.c``; built by ``oracle/cbuild.py`` = ``__graft_entry__.build()``, bound with ctypes), pinned against the
same recordings by ``tests/test_oracle_c.py``; it is the checker inside the torch-free C caller of the ABI
(``tests/c_abi/abi_host.c``).
"""
