"""CPU oracle for the B-frame codec hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product (``video-compression_amd/``)
never imports it and has no CPU fallback.

Contents
--------
``oracle.cai``      restatement of the CompressAI 1.1.8 pieces the reference path uses
                    (third-party, un-vendored: ``LHBDC/environment.yml:142``).
                    PARITY UNPINNED at that boundary: the real library is not installed,
                    the reference holds no golden vectors for it.
``oracle.lhbdc``    restatement of the reference's own wiring (``LHBDC/model/{m,flow,layers}.py``,
                    ``LHBDC/{encode_B,decode_B}.py``).  PINNED: ``oracle/gen_golden.py`` imports
                    the real reference modules in the build container (with ``oracle.cai`` as the
                    stand-in ``compressai``) and the fixtures under ``tests/golden/`` hold the
                    reference's outputs on seeded weights.
``oracle.flex``     same for ``Flex-Rate-Hier-Bidir-Video-Compression/b_model`` + ``test/``.
``oracle/rans_oracle.c``  plain-C rANS coder + pmf_to_quantized_cdf (``make -C oracle``).
"""
