"""The reference's training loop on the device: the fused kernel behind
`DeviceTrainer` against torch autograd, and a training run on the reference's
own rows (scans rebuilt from its metadata file, labels from its risk file)."""
import numpy as np
import pytest
import torch

from gym_auv_amd import trainer as T
from gym_auv_amd.models import LidarCNNShallow
from tests import scan_scenes as S
from tests.helpers import make_sim

pytestmark = pytest.mark.gpu
TOL = 1e-5   # fp32 parameters after a few dozen steps; summation orders differ from torch's


def _data(n, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.0, 1.0, (n, 1, 180)).astype(np.float32)
    X[rng.uniform(size=X.shape) < 0.6] = 0.0                # scans are mostly empty
    y = rng.uniform(0.0, 1.0, (n, 1)).astype(np.float32)
    return X, y


@pytest.mark.parametrize("optimizer,lr", [("adamw", 1e-3), ("sgd", 1e-2)])
@pytest.mark.parametrize("batch", [16, 5])
def test_fused_epochs_match_torch_autograd(optimizer, lr, batch):
    X, y = _data(83, 1)
    torch.manual_seed(4)
    ref_model = LidarCNNShallow()
    with torch.no_grad():   # make sure the output unit is alive at the start
        ref_model.linear[0].bias.fill_(0.3)
        ref_model.linear[0].weight.abs_()
    dev_model = LidarCNNShallow()
    dev_model.load_state_dict(ref_model.state_dict())
    ref = T.Trainer(ref_model, epochs=1, learning_rate=lr, optimizer=optimizer, device="cpu")
    dev = T.DeviceTrainer(dev_model, epochs=1, learning_rate=lr, X_train=X, y_train=y, optimizer=optimizer,
                          batch_size=batch)
    g = torch.Generator().manual_seed(0)
    for epoch in range(3):
        perm = torch.randperm(len(X), generator=g)
        got = dev.train_epoch(perm).cpu().numpy()
        want = []
        for i in range(len(X) // batch):
            j = perm[i * batch:(i + 1) * batch].numpy()
            want.append(ref.train_batch(torch.from_numpy(X[j]), torch.from_numpy(y[j])))
        assert np.abs(got - np.array(want)).max() <= TOL, epoch
    dev.sync_model()
    for k, v in ref_model.state_dict().items():
        assert np.abs(dev_model.state_dict()[k].numpy() - v.numpy()).max() <= TOL, k
    assert dev.steps_done == 3 * (len(X) // batch)
    assert np.mean(want) < 0.2


def test_training_on_the_reference_rows_reaches_the_published_error():
    """Scans of the reference's logged rows (rebuilt scenes through k_sense), labels = row-max risk, the
    reference's sequential split (utils/dataloader.py:63-67) and batch size (trainer.py:120-125): a freshly
    initialised one-convolution network trained by the fused kernel reaches the test error the reference
    publishes for that network (results_RL/_mse_1conv.txt: 0.0217 +- 0.0150)."""
    g = S.load("all")
    cfg, scene, label = S.build(g)
    sim = make_sim(cfg, scene)
    sim.reset(scene["init_state"])
    scans = sim.tensor("perception").clone()          # [n,1,180] on the device
    sim.close()
    tr, va, te = (torch.from_numpy(m) for m in S.split_masks(g))
    yy = torch.from_numpy(label.astype(np.float32))
    torch.manual_seed(2)                              # trainer.py:113
    model = LidarCNNShallow()
    t = T.DeviceTrainer(model, epochs=12, learning_rate=2e-3, X_train=scans[tr.cuda()], y_train=yy[tr],
                        X_val=scans[va.cuda()], y_val=yy[va], optimizer="adamw", batch_size=16, seed=2)
    train_loss, val_loss = t.train()
    assert t.steps_done == 12 * (int(tr.sum()) // 16) >= 3000
    assert train_loss[-1] < train_loss[0] and val_loss[-1] < 0.04
    from gym_auv_amd import cnn

    net = cnn.LidarCNN("shallow", model.state_dict())
    pred = net.forward(scans[te.cuda()].contiguous(), "risk")[:, 0].cpu().numpy()
    net.close()
    mse = float(np.mean((pred - label[te.numpy()]) ** 2))
    lo, hi = g["published_mse_mean"][0] - g["published_mse_std"][0], g["published_mse_mean"][0] + g["published_mse_std"][0]
    print("test MSE of the network trained here:", mse, "train/val loss per epoch:", train_loss, val_loss)
    assert lo < mse < hi, mse
