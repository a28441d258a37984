"""CPU oracle for the Context-Transformer detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may
be imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- never by ``context-transformer_amd/``.

Every function restates (in numpy / torch-CPU fp32, or plain C for the O(N^2)
NMS) the algorithm of one reference function and cites its ``file:line`` under
``/root/reference``.  The restatement is pinned against golden vectors that
``tools/gen_goldens.py`` produced by importing the reference itself in the
build container (``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``).

Parity status
-------------
* pinned by goldens: PriorBox, decode, encode, jaccard, point_form, match,
  Detect, box_utils.nms, py_cpu_nms (``>``), RFBNet-300 forward (phase 1,
  phase 2 transfer/incre, eval + train outputs), MultiBoxLoss_combined.
* pinned by known-answer + patched-in-/tmp Cython build: cpu_nms (``>=``).
* parity unpinned (reference raises IndexError): RFBNet-512 with the
  Context-Transformer block (models/RFB_Net_vgg.py:235-244).
"""
