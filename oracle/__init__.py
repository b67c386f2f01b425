"""CPU oracle for the DeepAVFusion/AVMAE pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  The product
path (``deepavfusion_amd``) never imports this package and fails loudly when
its HIP library is missing.

Parity status
-------------
* Everything the reference itself owns on this path (``models/deepavfusion.py``,
  ``models/fusion_blocks.py``, ``models/avmae.py``, ``models/vits.py``,
  ``util/pos_embed.py``, ``util/lr_sched.py``, ``util/misc.py`` grad-norm /
  step semantics) is PINNED: ``tests/golden/gen_golden.py`` imports those files from
  ``/root/reference`` in the build container and stores their outputs as
  fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
  restatement against them.
* The arithmetic that lives in un-vendored ``timm==0.9.2`` (``PatchEmbed``,
  ``Block``, ``Attention``, ``Mlp``; pinned in the reference's
  ``requirements.yml:23``) is NOT available here (no network).  It is restated
  from its published semantics and that part is
  "parity unpinned": the reference has no tests or golden vectors of its own.
"""
