"""CPU check of the fused HIP encoder's host half (csrc/encoder.hip, no GPU needed): the packed MFMA B-operand
image — ImageNet normalisation folded into layer 1, the padding-mask channel, the H/W axis swap, the k ordering of
the im2col fragments and the float16 hi/lo split — is decoded with numpy and pushed through a float64 emulation of
the kernel's data flow (padded RGB+mask input, conv / ReLU / max-pool stages as the kernel indexes them); the
result has to match the PyTorch CustomCNN evaluated on the reference's preprocessing."""
import numpy as np
import torch

from srlhip import _lib
from state_representation.models import SRLNeuralNetwork, preprocess

S1, S2 = 14, 36


def random_bn_net(state_dim, seed):
    torch.manual_seed(seed)
    net = SRLNeuralNetwork(state_dim, img_shape=(64, 64), backend="torch")
    for m in net.model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    return SRLNeuralNetwork(state_dim, img_shape=(64, 64), state_dict=net.model.state_dict(), backend="torch")


def decode_pack(pack, scales):
    """-> effective weights w1[64][224], w2[64][576], w3[64][576] with the per-layer pre-scales undone."""
    pack = pack.astype(np.float64)

    def layer(words, steps):
        frag = words.reshape(2, steps, 64, 16)                    # [n-half][k-step][lane][8 hi | 8 lo]
        eff = frag[..., :8] + frag[..., 8:]
        w = np.zeros((64, steps * 16))
        for nh in range(2):
            for lane in range(64):
                o, h = 32 * nh + (lane & 31), lane >> 5
                for s in range(steps):
                    w[o, 16 * s + 8 * h:16 * s + 8 * h + 8] = eff[nh, s, lane]
        return w

    n1, n2 = 2 * S1 * 64 * 16, 2 * S2 * 64 * 16
    return layer(pack[:n1], S1) / scales[0], layer(pack[n1:n1 + n2], S2) / scales[1], layer(pack[n1 + n2:], S2) / scales[2]


def emulate(img, w1, w2, w3, b2, b3, fcw, fcb):
    """float64 walk through the kernel's stages for ONE uint8 frame [64][64][3]."""
    inp = np.zeros((70, 72, 4))
    inp[3:67, 3:67, :3] = img
    inp[3:67, 3:67, 3] = 1.0
    c1 = np.zeros((32, 32, 64))
    for oy in range(32):
        for ox in range(32):
            patch = inp[2 * oy:2 * oy + 7, 2 * ox:2 * ox + 8, :].reshape(-1)      # k = ky*32 + slot*4 + c4
            c1[oy, ox] = w1 @ patch
    return emulate_from_c1(c1, w2, w3, b2, b3, fcw, fcb)


def emulate_from_c1(c1, w2, w3, b2, b3, fcw, fcb):
    """the walk from the raw conv-1 map [32][32][64] on"""
    c1 = np.maximum(c1, 0)
    pad = np.zeros((34, 34, 64)); pad[1:33, 1:33] = c1
    a2 = np.zeros((16, 16, 64))
    for py in range(16):
        for px in range(16):
            a2[py, px] = pad[2 * py:2 * py + 3, 2 * px:2 * px + 3].max(axis=(0, 1))
    pad = np.zeros((18, 18, 64)); pad[1:17, 1:17] = a2
    c2 = np.zeros((16, 16, 64))
    for oy in range(16):
        for ox in range(16):
            c2[oy, ox] = w2 @ pad[oy:oy + 3, ox:ox + 3, :].reshape(-1) + b2        # k = (ky*3 + kx)*64 + c
    c2 = np.maximum(c2, 0)
    a3 = np.zeros((7, 7, 64))
    for py in range(7):
        for px in range(7):
            a3[py, px] = c2[2 * py:2 * py + 3, 2 * px:2 * px + 3].max(axis=(0, 1))
    pad = np.zeros((9, 9, 64)); pad[1:8, 1:8] = a3
    c3 = np.zeros((4, 4, 64))
    for oy in range(4):
        for ox in range(4):
            c3[oy, ox] = w3 @ pad[2 * oy:2 * oy + 3, 2 * ox:2 * ox + 3, :].reshape(-1) + b3
    feat = np.maximum(c3, 0)[:3, :3].max(axis=(0, 1))
    return fcw @ feat + fcb


def test_pack_decodes_to_the_network():
    net = random_bn_net(5, 3)
    (w1, b1), (w2, b2), (w3, b3), (fw, fb) = net.folded_weights()
    pack, scales = _lib.encoder_pack(w1, b1, w2, w3)
    assert pack.nbytes == _lib.load().srlhip_encoder_pack_bytes() and np.isfinite(pack.astype(np.float32)).all()
    assert all(s > 0 and np.log2(s) == np.round(np.log2(s)) for s in scales)          # powers of two: undone exactly
    assert 8192 <= np.abs(pack.astype(np.float32)[:2 * S1 * 64 * 16]).max() <= 16384   # top of the scaled range
    e1, e2, e3 = decode_pack(pack, scales.astype(np.float64))
    # layers 2/3: plain weights with the two kernel axes swapped (the network sees the frame transposed)
    ref2 = np.transpose(w2, (0, 3, 2, 1)).reshape(64, 576)        # [o][ky][kx][c] with w[o][c][kx][ky]
    assert np.abs(e2 - ref2).max() <= 2.0 ** -21 * np.abs(ref2).max()
    ref3 = np.transpose(w3, (0, 3, 2, 1)).reshape(64, 576)
    assert np.abs(e3 - ref3).max() <= 2.0 ** -21 * np.abs(ref3).max()
    # layer 1: pixel slot 7 of every kernel row carries zero weights
    assert np.all(e1.reshape(64, 7, 8, 4)[:, :, 7, :] == 0)
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 256, size=(3, 64, 64, 3)).astype(np.uint8)
    imgs[1, :, :32] = 0                                            # flat regions: the padding-mask path matters at borders
    imgs[2] = 255
    with torch.no_grad():
        ref = net.model.getStates(preprocess(torch.from_numpy(imgs))).numpy()
    for i in range(3):
        out = emulate(imgs[i].astype(np.float64), e1, e2, e3, b2.astype(np.float64), b3.astype(np.float64),
                      fw.astype(np.float64), fb.astype(np.float64))
        assert np.abs(out - ref[i]).max() < 2e-4 * max(1.0, np.abs(ref[i]).max()), (i, out, ref[i])


def test_int8_layer1_pack_decodes_to_the_network():
    """Round 6: layer 1 of the fused kernel runs on the int8 matrix pipe (csrc/encoder.hip, pack_layer1_i8).  The digit image is
    decoded with numpy — balanced base-256 digits, per-channel power-of-two scales, slot 0 of every kernel row empty, mask tap around
    p - 128 — and the kernel's INTEGER data flow is emulated exactly (int64 sums of (p - 128) x digit, recombined as
    256 a2 + a1 + floor(a0 / 256)): followed by the float64 walk through layers 2 / 3 / FC it has to match PyTorch."""
    net = random_bn_net(5, 3)
    (w1, b1), (w2, b2), (w3, b3), (fw, fb) = net.folded_weights()
    dig, inv = _lib.encoder_pack_i8(w1, b1)
    assert dig.shape == (2, 7, 3, 64, 16) and dig.dtype == np.int8
    assert all(v > 0 and np.log2(v) == np.round(np.log2(v)) for v in inv)                  # powers of two: undone exactly
    # [o][k = row * 32 + slot * 4 + c] per digit
    z = np.zeros((3, 64, 224), np.int64)
    for nh in range(2):
        for lane in range(64):
            o, h = 32 * nh + (lane & 31), lane >> 5
            for s in range(7):
                for d in range(3):
                    z[d, o, 32 * s + 16 * h:32 * s + 16 * h + 16] = dig[nh, s, d, lane]
    fixed = 65536 * z[0] + 256 * z[1] + z[2]
    assert np.abs(fixed).max(axis=1).min() > 2 ** 21 and np.abs(fixed).max() <= 127 * 65536 + 127 * 256 + 127     # every channel uses its 24 bits
    assert np.all(fixed.reshape(64, 7, 8, 4)[:, :, 0, :] == 0)                             # the fragment's first pixel slot carries no tap
    # effective real weights: colour taps multiply p - 128, the mask tap carries the rest
    eff = fixed.reshape(64, 7, 8, 4) * (inv.astype(np.float64) / 256.0)[:, None, None, None]
    std, mean = np.array([0.229, 0.224, 0.225], np.float32).astype(np.float64), np.array([0.485, 0.456, 0.406], np.float32).astype(np.float64)   # (the packer's float constants)
    wt = np.transpose(w1.astype(np.float64), (0, 3, 2, 1))                                  # [o][ky][kx][c] = w[o][c][kx][ky] (the frame is seen transposed)
    step = (inv.astype(np.float64) / 256.0)[:, None, None, None]                           # one fixed-point unit of channel o
    assert np.all(np.abs(eff[:, :, 1:, :3] - wt / (255.0 * std)) <= 0.501 * step)
    assert np.all(step[:, 0, 0, 0] <= 2.0 ** -21 * np.abs(eff).reshape(64, -1).max(axis=1))  # >= 22 bits below the channel's largest tap
    mask_ref = (wt * ((128.0 / 255.0 - mean) / std)).sum(axis=3)
    mask_ref[:, 3, 3] += b1
    mask_ref /= 127.0                                                                       # the mask byte of an inside pixel is 127
    assert np.all(np.abs(eff[:, :, 1:, 3] - mask_ref) <= 0.501 * step[:, :, :, 0])
    pack, scales = _lib.encoder_pack(w1, b1, w2, w3)
    _, e2, e3 = decode_pack(pack, scales.astype(np.float64))
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 256, size=(3, 64, 64, 3)).astype(np.uint8)
    imgs[1, :, :32] = 0
    imgs[2] = 255
    with torch.no_grad():
        ref = net.model.getStates(preprocess(torch.from_numpy(imgs))).numpy()
    for i in range(3):
        inp = np.zeros((70, 72, 4), np.int64)                                              # pixel x at slot x + 4, 3 rows of top margin
        inp[3:67, 4:68, :3] = imgs[i].astype(np.int64) - 128
        inp[3:67, 4:68, 3] = 127
        c1 = np.zeros((32, 32, 64))
        for oy in range(32):
            for ox in range(32):
                patch = inp[2 * oy:2 * oy + 7, 2 * ox:2 * ox + 8, :].reshape(-1)
                a = z @ patch                                                              # [3][64] exact digit sums
                assert np.abs(a).max() < 2 ** 22
                comb = 256 * a[0] + a[1] + (a[2] >> 8)
                assert np.abs(comb).max() < 2 ** 30
                c1[oy, ox] = comb.astype(np.float64) * inv
        out = emulate_from_c1(c1, e2, e3, b2.astype(np.float64), b3.astype(np.float64), fw.astype(np.float64), fb.astype(np.float64))
        assert np.abs(out - ref[i]).max() < 2e-4 * max(1.0, np.abs(ref[i]).max()), (i, out, ref[i])


def test_pack_rejects_short_buffers_and_encoder_needs_a_gpu():
    lib = _lib.load()
    z = np.zeros(16, np.float32)
    assert lib.srlhip_encoder_pack(_lib._ptr(z), _lib._ptr(z), _lib._ptr(z), _lib._ptr(z), _lib._ptr(z), 16, _lib._ptr(z)) == -22
    assert _lib.encoder_supported(224, 224, 3) and _lib.encoder_supported(64, 64, 6)      # the layered path (encoder_general.hip)
    assert not _lib.encoder_supported(64, 64, 4) and not _lib.encoder_supported(32, 32, 3)
    if not torch.cuda.is_available():
        net = SRLNeuralNetwork(2, img_shape=(64, 64))              # CPU device: PyTorch forward, never the HIP handle
        assert net.backend == "torch" and net.hip is None
