"""CPU oracle: a NumPy (fp32) restatement of the FreeRL hot path.

TEST INFRASTRUCTURE — NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this package, and there only as the checker /
the timed CPU baseline.  `freerl_amd/` never imports it and has no CPU fallback: without
the HIP library it raises.

Parity status: PINNED.  The reference publishes no golden vectors or tests,
so the oracle is pinned against outputs of the reference itself, produced in the build
container by importing its classes (`tests/golden/make_golden.py`) on seeded synthetic
inputs; `tests/test_oracle_golden.py` checks every function here against those fixtures.
The third-party arithmetic on the path (torch `nn.Linear`/autograd/`optim.Adam`/
`clip_grad_norm_`/`distributions.Normal`, versions per reference README.md:23-35) is restated
from its published semantics; the call sites are cited per function.

Every function follows a cited reference file:line; all arithmetic is float32 like the
reference's tensors (Python/NumPy scalars only where the reference uses them).
"""
from . import nn, buffer, algos, ppo, normalization  # noqa: F401
