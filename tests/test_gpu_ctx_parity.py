"""Context-Transformer parity at the batch sizes BASELINE.json names (configs[2]: bs 32), not only bs 2:
bs {2, 8, 32} x seeds {1234, 7, 99}, EVERY element of the block's output on the oracle's image subset, against the
reference's fp32 CPU arithmetic AND an fp64 evaluation (models/RFB_Net_vgg.py:253-271).

What is asserted (tools/ctx_parity.py --budget, profiles/r04_ctx_parity.txt):
  * the block's INPUT (raw conf-head output) and loc / obj: 1e-4 vs the CPU fp32 path (measured: 2e-6);
  * the block's own arithmetic on an identical input: 1e-4 (measured 1.0e-5; torch-CPU fp32's is 1.6e-5);
  * the composite: 1e-4 vs the CPU fp32 path, every element, no second clause (ctx_cases.verdict; the fp64 block amplifies a
    perturbation of its input ~1000x, the CPU path itself is 4.8..7.2e-5 from fp64 and moves inside that band with the host's
    thread count; conftest.py pins 8 threads) -- against the reference at 8 threads AND, same device output, at 128
    (profiles/r06_ctx_policy.txt: the sweep that chose the shipped policy).
Inputs are 'randn' (SURVEY 8d (i)).  On image-like 'u8' inputs (8d (ii), |x| ~ 128) the logits are ~1e4 and the
block is chaotic in fp32: torch-CPU fp32 itself is 1e-3 .. 1e-1 away from fp64 there, so no fp32 implementation has a
parity to meet; that case only checks the block's input.  The shipped kernel policy (engine.ctx_policy 'h2': the committed table
with its F(4x4,3x3) entries on the f16x2 operand form, direct layers on bf16x3 with two accumulators) runs here."""
import pytest
import torch

import ctx_cases as cc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def net300():
    net = cc.build(300, 60)
    return net, cc.state(net), cc.state(net, torch.float64)


@pytest.mark.parametrize('batch', [2, 8, 32])
@pytest.mark.parametrize('seed', [1234, 7, 99])
def test_phase2_parity_sweep(net300, batch, seed):
    net, sd32, sd64 = net300
    r = cc.sweep_case(net, 300, 60, 'transfer', batch, seed, 'randn', sd32, sd64, extra_threads=(128,))
    assert r['loc_gpu_cpu32'] < 1e-4 and r['obj_gpu_cpu32'] < 1e-4 and r['rawconf_gpu_cpu32'] < 1e-4, r
    assert cc.verdict(r) == 'ok', r
    # the same device output against the fp32 CPU reference evaluated with 128 threads (torch's CPU convolutions split their sums
    # by thread: another, equally valid fp32 reference)
    assert r['gpu_cpu32@128'] <= 1e-4, r


def test_phase2_parity_512_at_the_per_gpu_batch_of_configs3():
    """BASELINE configs[3] = RFBNet-512 + Context-Transformer at bs 64 over 8 GPUs: the per-GPU shape (bs 8; M = 4 964
    context rows, 32 756 priors: the longest softmax rows of any configuration).  Everything up to the block's input
    holds 1e-4 against the CPU fp32 path.  The block's output has no 1e-4 parity between ANY two fp32 evaluations here:
    with the synthetic weights the un-scaled logits are ~400 and torch-CPU fp32 is itself 8e-4 from the fp64
    evaluation (measured: CPU32-fp64 7.9e-4, GPU-fp64 8.1e-4, GPU-CPU32 9.1e-4), so the device is held to the truth
    instead: no further from fp64 than 1.5 x the CPU path in the max norm and 1.75 x at the 99.99 % quantile."""
    net = cc.build(512, 60)
    r = cc.sweep_case(net, 512, 60, 'transfer', 8, 1234, 'randn')
    assert r['loc_gpu_cpu32'] < 1e-4 and r['obj_gpu_cpu32'] < 1e-4 and r['rawconf_gpu_cpu32'] < 1e-4, r
    assert r['gpu_fp64'] <= max(1e-4, 1.5 * r['cpu32_fp64']), r
    assert r['q_gpu_fp64'] <= max(1e-4, 1.75 * r['q_cpu32_fp64']), r


@pytest.mark.parametrize('batch', [2, 32])
def test_phase2_image_like_input_block_input_parity(net300, batch):
    """SURVEY 8d (ii) inputs: everything up to the block's input holds 1e-4; the block's output has no fp32 parity
    there (the CPU fp32 path is 1e-3 .. 1e-1 from fp64), which this test records instead of hiding."""
    net, sd32, sd64 = net300
    r = cc.sweep_case(net, 300, 60, 'transfer', batch, 1234, 'u8', sd32, sd64)
    assert r['loc_gpu_cpu32'] < 1e-4 and r['obj_gpu_cpu32'] < 1e-4 and r['rawconf_gpu_cpu32'] < 1e-4, r
    assert r['cpu32_fp64'] > 1e-3, r           # if this ever fails the randn-only restriction above can go


def test_budget_upstream_dominates(net300):
    """The error budget that justifies the criterion: the block's own fp32 arithmetic (device kernel on an exactly
    rounded input) is several times smaller than what the block makes of the fp32 rounding of its input."""
    net = net300[0]
    rows = dict(cc.budget(net, 300, 60, 2))
    kern = rows['  block only     (fp64 conf rounded to fp32 -> GPU attention kernel)']
    cpu_kern = rows['  block only     (same input -> torch-CPU fp32 block)']
    up = rows['  upstream only  (GPU conf+pool -> fp64 block)']
    assert kern < 1e-4 and kern <= 2.0 * cpu_kern + 1e-6, rows
    assert up > kern, rows
