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
