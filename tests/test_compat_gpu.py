"""The reference's RL-side extractors (radarCNN.py:10-280) on the GPU: frozen
and in their shipped configuration they run through the library's CNN kernel,
on the simulation's perception buffer in place; anything else (gradients
wanted, other channel counts) takes the torch layers."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-6   # float32 sums of up to 45 products, fused in the kernel and not in torch


def _state(tag):
    import torch

    gold = np.load(os.path.join(G, "cnn_golden.npz"))
    pre = f"w_{tag}/"
    return gold, {k[len(pre):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,cls,dim", [("shallow", "LidarCNN_shallow_pretrained", 12),
                                         ("deep2", "LidarCNN_not_so_deep_pretrained", 12),
                                         ("deep", "LidarCNN_deep_pretrained", 8)])
def test_frozen_extractor_runs_through_the_kernel_and_equals_its_torch_layers(tag, cls, dim):
    import torch

    from gym_auv_amd import compat
    from gym_auv_amd.env import Box

    gold, shipped = _state(tag)
    net = getattr(compat, cls)(observation_space=Box(0.0, 1.0, (1, 180)), features_dim=dim)
    net.load_state_dict({k: v for k, v in shipped.items() if k in net.state_dict()})
    net = net.cuda()
    x = torch.from_numpy(gold["x"]).cuda()
    y_train = net(x)                                   # parameters still ask for gradients: the torch layers
    assert y_train.requires_grad and getattr(net, "_net", None) is None
    for p in net.parameters():
        p.requires_grad = False
    y = net(x)
    assert getattr(net, "_net", None) is not None and not y.requires_grad
    assert y.shape == (24, dim)
    assert np.abs(y.cpu().numpy() - gold[f"{tag}_features"]).max() <= TOL
    assert np.abs(y.cpu().numpy() - y_train.detach().cpu().numpy()).max() <= TOL
    # new weights are seen: the handle is rebuilt
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    first = net._net
    y2 = net(x)
    assert net._net is not first
    ref = x
    for layer in net._layers():
        ref = layer(ref)
    assert np.abs(y2.cpu().numpy() - ref.cpu().numpy()).max() <= TOL


def test_extractor_reads_the_simulations_observation_in_place(tmp_path):
    import torch

    from gym_auv_amd import compat, scenarios
    from gym_auv_amd.env import VecAuvEnv

    gold, shipped = _state("shallow")
    path = str(tmp_path / "model_shallow_pretrained.json")
    torch.save(shipped, path)
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=512)
    env = VecAuvEnv(cfg, scene)
    ext = compat.PerceptionNavigationExtractor(env.observation_space, cnn_path=path).cuda()
    obs = env.reset()
    feats = ext(obs)
    assert feats.shape == (512, 18) and feats.is_cuda
    conv = ext.extractors["perception"]
    want = obs["perception"]
    for layer in conv._layers():
        want = layer(want)
    assert np.abs(feats[:, :12].cpu().numpy() - want.cpu().numpy()).max() <= TOL
    assert torch.equal(feats[:, 12:], obs["navigation"][:, 0, :])
    env.close()
