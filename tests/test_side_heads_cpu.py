"""The optional heads of LSEField (R:lse_nerf/lse_field.py:190-252, 313-345; off in every LSENeRF preset): their modules are
plain torch on whatever device the tensors live on, so parameter layout, names and arithmetic are held against the oracle here,
without a GPU.  (Parity unpinned: the reference holds no fixture for these heads either.)"""
import math

import torch

from lsenerf_amd.field import DenseMLP, FieldHead, FieldHeadNames, FrequencyEncoding, PredNormalsFieldHead
from oracle.field import SideHeadsOracle, TcnnMLP, frequency_encoding_tcnn


def test_dense_mlp_is_a_tcnn_layout_network():
    for (i, layers, w, o) in ((31, 2, 64, 64), (15, 2, 64, 64), (27, 3, 64, 64), (15, 2, 32, 40)):
        torch.manual_seed(i)
        m = DenseMLP(in_dim=i, num_layers=layers, layer_width=w, out_dim=o)
        ref = TcnnMLP(i, layers, w, o, None)
        assert m.params.numel() == ref.n_params and m.shapes == ref.shapes
        x = torch.randn(257, i, dtype=torch.float64)
        p = m.params.detach().double().requires_grad_(True)
        y_ref = ref.forward(x, p)
        xm = x.float().requires_grad_(True)
        y = m(xm)
        assert y.shape == (257, o)
        assert (y.double() - y_ref).abs().max() < 1e-5
        g = torch.randn_like(y)
        y.backward(g)
        y_ref.backward(g.double())
        assert (m.params.grad.double() - p.grad).abs().max() < 1e-4 * max(1.0, p.grad.abs().max().item())
        # leading shape is kept
        assert m(torch.randn(3, 5, i)).shape == (3, 5, o)


def test_frequency_encoding_is_tcnn_frequency():
    enc = FrequencyEncoding(in_dim=3, num_frequencies=2)
    assert enc.get_out_dim() == 12
    x = torch.rand(101, 3, dtype=torch.float64) * 4 - 2
    assert (enc(x) - frequency_encoding_tcnn(x, 2)).abs().max() < 1e-12
    # spelled out for one point: per input dimension (sin, cos) of pi x, then (sin, cos) of 2 pi x
    one = enc(torch.tensor([[0.25, 0.0, -0.5]], dtype=torch.float64))[0]
    want = []
    for v in (0.25, 0.0, -0.5):
        for e in (0, 1):
            want += [math.sin(v * 2 ** e * math.pi), math.cos(v * 2 ** e * math.pi)]
    assert (one - torch.tensor(want, dtype=torch.float64)).abs().max() < 1e-12
    assert FrequencyEncoding(2, 3)(torch.zeros(4, 2)).shape == (4, 12)


def test_field_heads_have_nerfstudio_names_and_activations():
    h = FieldHead(5, FieldHeadNames.SEMANTICS, 64, None)
    assert sorted(h.state_dict()) == ["net.bias", "net.weight"] and h.net.weight.shape == (5, 64)
    x = torch.randn(9, 64)
    assert torch.equal(h(x), h.net(x))
    sp = FieldHead(1, FieldHeadNames.UNCERTAINTY, 64, torch.nn.Softplus())
    assert (sp(x) > 0).all()
    n = PredNormalsFieldHead(64)(x)
    assert n.shape == (9, 3) and (n.norm(dim=-1) - 1).abs().max() < 1e-6


def test_side_heads_oracle_runs_on_reference_named_parameters():
    torch.manual_seed(0)
    geo, tdim, hid = 15, 16, 64
    p = {"embedding_transient.embedding.weight": torch.randn(7, tdim),
         "mlp_transient.params": torch.randn(TcnnMLP(geo + tdim, 2, hid, hid, None).n_params) * 0.1,
         "mlp_semantics.params": torch.randn(TcnnMLP(geo, 2, 64, hid, None).n_params) * 0.1,
         "mlp_pred_normals.params": torch.randn(TcnnMLP(geo + 12, 3, 64, hid, None).n_params) * 0.1}
    for name, o in (("field_head_transient_uncertainty", 1), ("field_head_transient_rgb", 3), ("field_head_transient_density", 1),
                    ("field_head_semantics", 11), ("field_head_pred_normals", 3)):
        p[name + ".net.weight"], p[name + ".net.bias"] = torch.randn(o, hid) * 0.1, torch.randn(o) * 0.1
    orc = SideHeadsOracle(p)
    g = torch.randn(33, geo, requires_grad=True)
    u, rgb, dens = orc.transient(g, torch.randint(0, 7, (33, 1)))
    assert u.shape == (33, 1) and rgb.shape == (33, 3) and dens.shape == (33, 1) and (u > 0).all() and (dens > 0).all()
    sem = orc.semantics(g)
    assert sem.shape == (33, 11) and not sem.requires_grad            # pass_semantic_gradients=False detaches the input
    assert orc.semantics(g, pass_gradients=True).requires_grad
    nrm = orc.pred_normals(torch.randn(33, 3), g)
    assert (nrm.norm(dim=-1) - 1).abs().max() < 1e-6
