"""GPU (MI355X): backward kernels, called through the C ABI, against torch autograd (fp64, CPU) of the
same op.  dgrad runs the forward implicit-GEMM kernel on flipped/transposed weights (bf16x3), wgrad the
exact-fp32 MFMA kernel with deterministic split-K partials."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 5e-5


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def lib(built_lib):
    from tedm_amd import _lib
    return _lib.lib()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CONV_BWD_CASES = [
    # k, stride, ups, Cin, Cout, H, W, B
    (3, 1, 0, 64, 64, 24, 40, 2),
    (3, 1, 0, 96, 128, 17, 33, 2),
    (3, 1, 1, 128, 64, 12, 20, 2),
    (4, 2, 0, 64, 128, 32, 48, 2),
    (1, 1, 0, 128, 384, 9, 40, 3),
    (1, 1, 0, 192, 64, 16, 16, 2),
]


@pytest.mark.parametrize('case', CONV_BWD_CASES, ids=lambda c: 'k{}s{}u{}_{}to{}_{}x{}'.format(*c[:7]))
def test_conv_backward(lib, dev, case):
    from tedm_amd import _lib
    k, stride, ups, Cin, Cout, H, W, B = case
    x = rnd(B, Cin, H, W, seed=1).double().requires_grad_(True)
    w = (rnd(Cout, Cin, k, k, seed=2) / (Cin * k * k) ** 0.5).double().requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    y = F.conv2d(xin, w, None, stride=stride, padding={3: 1, 4: 1, 1: 0}[k])
    dy = rnd(*y.shape, seed=3).double()
    y.backward(dy)
    d = lambda t: t.float().contiguous().to(dev)
    DY, X, Wt = d(nhwc(dy)), d(nhwc(x.detach())), d(w.detach())
    outs = []
    for _ in range(2):
        dx = torch.full((B, H, W, Cin), float('nan'), device=dev)
        dw = torch.full((Cout, Cin, k, k), float('nan'), device=dev)
        _lib.check(lib.tedm_op_conv2d_backward(DY.data_ptr(), X.data_ptr(), Wt.data_ptr(), B, H, W, Cin, Cout, k, stride, ups,
                                               dx.data_ptr(), dw.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream),
                   'tedm_op_conv2d_backward')
        outs.append((dx.permute(0, 3, 1, 2).cpu(), dw.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # deterministic
    assert rel(outs[0][0], x.grad) < TOL
    assert rel(outs[0][1], w.grad) < TOL
