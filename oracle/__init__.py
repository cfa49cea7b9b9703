"""CPU oracle for the ComMU Transformer-XL hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker / reported baseline, never as the thing
measured or shipped.  The product path (``commu-code_amd/``) never imports it and
fails loudly when the HIP library is missing.

What it is: a plain PyTorch fp32 restatement (functional, index-math form) of the
reference's algorithm for the path SURVEY.md section 8 names -- the forward/backward
of ``MemTransformerLM`` (reference ``commu/model/model.py``), the ``train()`` inner
step (reference ``train.py:128-169``) and the sampling step / decode loop
(reference ``commu/midi_generator/midi_inferrer.py:186-320``).

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: ``tests/golden/make_golden.py`` imports ``/root/reference`` in the build
container and writes the fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every oracle function against them
(<= 1e-5 abs on fp32 values, exact on integers).
"""
