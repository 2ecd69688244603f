"""fp32 engine against the float64 oracle AT THE QUOTED SIZES (VERDICT r3 "what's weak" #2): the whole stage-1 step of
BASELINE configs[1] at its batch of 256 stamps, and the 128 x 128 x 6 / six-level net of configs[3] at its per-GPU batch
of 64 - the persistent multi-item paths of the Winograd kernels (several items per workgroup, the LDS ring across item
boundaries), the split ranges of the Winograd weight gradient, the 870-block BN statistics pass, the 384-workgroup tiled
weight gradients.  Same checks and tolerances as tests/test_gpu_parity.py::_run_parity (outputs 2e-4 * max, ELBO 1e-4
relative, Adam update against the oracle's), both keep_outputs forms, one train step.  Gradients: 1e-3 * max per tensor,
or - per tensor - no worse than 1.5 x what a numpy float32 evaluation of the same step misses float64 by, capped at 1e-2
(_run_parity's f32_floor: at these sizes float32 itself is up to 2e-2 * max away from float64 on the early encoder tensors
where the engine is at 1.5e-3; the per-layer bound of 2e-5 is tests/test_gpu_0_layers_f32.py's).
Reference semantics: training/train.py:27-37 (fit's train_function at the BASELINE batch), model.py:61-161.

This file sorts in front of the HIP-vs-HIP kernel cross-checks (test_gpu_parity.py) so that `-x` reaches the oracle first.

Two parameter sets at 256 stamps: the Keras-like initialisation of every other parity test (sigma sits on its 1e-4 floor
wherever the head's relu is closed: 1 / sigma^2 = 1e8 multiplies every rounding of the mean) and the same with the head's
scale bias shifted by +0.3 (sigma ~ 0.3: a conditioning in which a wrong tap in a low-energy layer cannot hide behind the
floor pixels' 1e8 weights).
"""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from tests.test_gpu_parity import _run_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sigma_bias", [0.0, 0.3])
def test_full_arch_stage1_step_at_batch_256_against_the_oracle(sigma_bias):
    # (sigma on its floor: also against the float64 oracle at the engine's own gate states - 1e-3 on every tensor, see the
    # 128-px test below; the second parameter set keeps the float32-floor rule only, to bound the suite's run time)
    worst = _run_parity(vo.Arch(), B=256, seed=2, data_seed=5, sigma_bias=sigma_bias, f32_floor=True,
                        gate_matched=sigma_bias == 0.0)
    print(f"\n59 px, 256 stamps, sigma bias {sigma_bias}: largest gradient error {worst[1]:.2e} * max ({worst[0]})")


def test_full_arch_stage2_frozen_decoder_at_batch_256():
    # stage 2 of train_deblender (train.py:175-183): decoder frozen, gradients still flow through it
    _run_parity(vo.Arch(), B=256, seed=3, data_seed=6, train_decoder=False, sigma_bias=0.3, f32_floor=True)


def test_128px_six_level_arch_at_its_per_gpu_batch_of_64():
    """VERDICT r5 "what's weak" 2 / item 4(i): on this case the engine's distance from float64 equalled the numpy-float32
    evaluation's to four digits on seven decoder tensors (dec/prelu_in/alpha 3.635e-3 vs 3.636e-3 ...) - one discrete event
    common to both, not "float32's noise".  The event (tests/probe_f32_vs_f64*.py, profiles/r06_gate_flip_probe.txt): PReLU
    GATES.  1628 of the 2.5e8 pre-activations of this step lie within float32 rounding of zero and come out on the other
    side in a float32 forward; the backward multiplies by 1 or by alpha there.  For the seven decoder tensors it is the gate
    of decoder layer 2 at stamp 51, pixel (5, 4), channel 239, whose pre-activation is +1.7e-9 in float64 and <= 0 in
    float32 - in numpy's evaluation and (a coin that fell the same way) in the engine's.  Imposing ONLY the float32 gate
    states on the float64 evaluation reproduces the whole float32-vs-float64 distance (3.98e-2 on enc/prelu0/alpha) and
    leaves 2.9e-4 * max as the largest remainder.  So the test now ALSO compares with the float64 oracle evaluated at the
    engine's own gate states (gate_matched): every tensor within the file header's 1e-3 (2e-3 BatchNorm), no exception."""
    arch = vo.Arch(input_shape=(128, 128, 6), latent_dim=32, filters=(32, 64, 128, 256, 512, 512), kernels=(3,) * 6)
    worst = _run_parity(arch, B=64, seed=21, sigma_bias=0.3, f32_floor=True, gate_matched=True)
    print(f"\n128 px, 64 stamps: largest gradient error {worst[1]:.2e} * max ({worst[0]})")
