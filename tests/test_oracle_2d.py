"""The functional 2D oracle must agree with an nn.Module composition of the same topology (wiring check, CPU)."""
import numpy as np
import torch

from oracle.net2d_ref import net2d_forward


def test_oracle2d_matches_module_composition_eval_and_shapes():
    from mm2d3d_amd.net2d import Net2DSeg

    torch.manual_seed(0)
    net = Net2DSeg(6, pretrained=False).eval()
    g = np.random.default_rng(0)
    B, H, W = 2, 30, 44  # padded to 32x48 inside
    idx = [np.stack([g.integers(0, H, 50), g.integers(0, W, 50)], 1) for _ in range(B)]
    batch = {"img": torch.rand(B, 3, H, W), "depth": torch.rand(B, 1, H, W), "img_indices": idx}
    sd = {k: v for k, v in net.state_dict().items()}
    preds, last, _, aux = net2d_forward(sd, batch, training=False)
    assert preds["seg_logit"].shape == (100, 6) and preds["seg_logit_2d"].shape == (B, 6, H, W)
    assert last.shape == (B, 64, H, W) and aux["seg_logit_avg"].shape == (100, 6)
    # module composition on CPU: only the lifting op needs the GPU, so compare the dense maps
    with torch.no_grad():
        r = net.rgb_backbone(torch.nn.functional.pad(batch["img"], [0, 4, 0, 2]))
    from oracle.net2d_ref import backbone

    rr = backbone(sd, "rgb_backbone", torch.nn.functional.pad(batch["img"], [0, 4, 0, 2]), False, None)
    for a, b in zip(r, rr):
        assert torch.allclose(a, b, atol=1e-5)
    # gather semantics: point j of sample i reads pixel (row, col)
    i, j = 1, 7
    rr_, cc_ = idx[i][j]
    assert torch.equal(preds["seg_logit"][50 * i + j], preds["seg_logit_2d"][i, :, rr_, cc_])
