"""CPU oracle for the DEP-GAN two-critic WGAN-GP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``dep_gan_im_amd``) never imports this package and fails loudly when the HIP
library is missing.

PARITY UNPINNED: the reference (Keras 2.x / TF 1.x, Python 2) cannot run in
this environment and ships no golden vectors, tests or weights, so this
restatement is pinned only by (a) two independent implementations agreeing
(autograd graph in ``depgan_oracle.py`` vs hand-derived backward in
``manual.py``) and (b) fixtures generated from it under ``tests/golden``.
"""
