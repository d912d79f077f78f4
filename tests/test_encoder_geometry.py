"""Host-only checks of the layered encoder's shape logic (csrc/encoder_general.hip::geometry, through the C-ABI): which
frame shapes the HIP encoder covers and how many inputs its fully connected layer has, against PyTorch's own shape
arithmetic for the same CustomCNN (state_representation/models.py; reference: models.py:178-193)."""
import pytest
import torch

from srlhip import _lib
from state_representation.models import CustomCNN


@pytest.mark.parametrize("shape", [(64, 64), (224, 224), (96, 128), (128, 96), (75, 61), (48, 56), (41, 41), (200, 320), (1024, 48)])
@pytest.mark.parametrize("ch", [3, 6])
def test_feature_count_is_torchs_flatten_size(shape, ch):
    with torch.no_grad():
        net = CustomCNN(2, ch, shape)
    assert _lib.encoder_supported(shape[0], shape[1], ch)
    assert _lib.encoder_feature_count(shape[0], shape[1], ch) == net.flat_dim


def test_uncovered_shapes():
    for h, w, c in ((64, 64, 4), (64, 64, 1), (40, 40, 3), (64, 40, 3), (7, 224, 3), (1025, 64, 3), (64, 2048, 6)):
        assert not _lib.encoder_supported(h, w, c) and _lib.encoder_feature_count(h, w, c) == 0
    # the smallest frames the three conv + pool stages leave one cell of: 41x41 (torch refuses 40x40 as well)
    assert _lib.encoder_feature_count(41, 41, 3) == 64 and _lib.encoder_feature_count(40, 40, 3) == 0
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            CustomCNN(2, 3, (40, 40))


@pytest.mark.parametrize("ch", [3, 6])
def test_layered_layer1_pack_is_the_normalised_convolution(ch):
    """The layered path's layer-1 B image (srlhip_encoder_pack_first_layer, csrc/encoder_general.hip) decoded with numpy and applied
    to a padded (channels, mask[, 0]) uint8 frame the way the kernel indexes it — k = (ky * 8 + kx) * cpix + c, stride 2,
    3-pixel ring with mask 0 — equals the BatchNorm-folded conv1 of the torch model on the reference's preprocessing
    (scale to [0, 1], ImageNet mean / std per 3-channel group, H/W swapped): checks the folding, the mask channel that keeps
    zero padding exact in normalised space, the axis swap and the k order for 3- and 6-channel frames, without a GPU."""
    import numpy as np
    from state_representation.models import SRLNeuralNetwork, preprocess
    torch.manual_seed(4 + ch)
    H, W = 22, 30                                           # not square: an axis mix-up cannot hide
    net = SRLNeuralNetwork(3, img_shape=(64, 64), n_channels=ch, backend="torch")
    for m in net.model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    net = SRLNeuralNetwork(3, img_shape=(64, 64), n_channels=ch, state_dict=net.model.state_dict(), backend="torch")
    (w1, b1) = net.folded_weights()[0]
    pack, scale = _lib.encoder_pack_first_layer(w1, b1)
    assert scale > 0 and np.log2(scale) == np.round(np.log2(scale))
    cpix, ks = (4, 14) if ch == 3 else (8, 28)
    frag = pack.astype(np.float64).reshape(2, ks, 64, 16)       # [channel half][k-step][lane][8 hi | 8 lo]
    eff = (frag[..., :8] + frag[..., 8:]) / scale
    wk = np.zeros((64, ks * 16))
    for nh in range(2):
        for lane in range(64):
            for s in range(ks):
                wk[32 * nh + (lane & 31), 16 * s + 8 * (lane >> 5):16 * s + 8 * (lane >> 5) + 8] = eff[nh, s, lane]
    assert np.all(wk.reshape(64, 7, 8, cpix)[:, :, 7, :] == 0)                         # pixel slot 7 carries no weight
    if ch == 6:
        assert np.all(wk.reshape(64, 7, 8, cpix)[..., 7] == 0)                         # nor does the fill channel
    rs = np.random.RandomState(ch)
    img = rs.randint(0, 256, size=(H, W, ch)).astype(np.uint8)
    img[:, :7] = 0                                          # a flat border: the padding-mask path matters there
    inp = np.zeros((H + 6, W + 8, cpix))
    inp[3:3 + H, 3:3 + W, :ch] = img
    inp[3:3 + H, 3:3 + W, ch] = 1.0
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = np.zeros((Hc, Wc, 64))
    for oy in range(Hc):
        for ox in range(Wc):
            out[oy, ox] = wk @ inp[2 * oy:2 * oy + 7, 2 * ox:2 * ox + 8, :].reshape(-1)
    conv1 = [m for m in net.fused_conv if isinstance(m, torch.nn.Conv2d)][0]
    with torch.no_grad():
        ref = conv1(preprocess(torch.from_numpy(img[None]))).numpy()[0]                # [64][W'][H']: the network sees W and H swapped
    ref = np.transpose(ref, (2, 1, 0))                                                 # -> [frame row][frame column][channel]
    assert ref.shape == out.shape
    assert np.abs(out - ref).max() < 2e-5 * np.abs(ref).max()
