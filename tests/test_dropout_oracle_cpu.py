"""CPU checks of the dropout oracle (oracle/dropout.py): the numpy Philox core against the Random123 known answers and
the C oracle, the mask definition's structure (every hidden unit gets exactly one draw, kept at the stated rate,
independent across rows / steps / layers / seeds / kernels), the masked forward pass against torch's own nn.Dropout
arithmetic, and -- on the reference's trained network and paths -- that the oracle alone reproduces what dropout at
inference does to the reference's price (tests/golden/v3_frozen_nn.npz: 7.2142 in eval mode, 7.0225 as the reference
runs it)."""
import numpy as np
import pytest

from oracle import cpu as ocpu
from oracle import dropout as dr
from oracle import reference_flow as rf


def test_numpy_philox_matches_the_known_answers_and_the_c_oracle(golden):
    kats = golden["scalars"].get("philox_kat")
    cases = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
             ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
             ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
              (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in cases:
        got = dr.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    assert kats is None or len(kats) >= 1  # the same vectors pin the C oracle (test_oracle_golden.py)
    rng = np.random.default_rng(1)
    ctr = rng.integers(0, 2 ** 32, (200, 4), dtype=np.uint64)
    k0, k1 = 0x9ABCDEF0, 0x12345678
    got = np.stack(dr.philox4x32_10(ctr[:, 0], ctr[:, 1], ctr[:, 2], ctr[:, 3], k0, k1), axis=1)
    for i in range(200):
        assert np.array_equal(got[i], ocpu.philox4x32_10(ctr[i].astype(np.uint32), np.array([k0, k1], np.uint32)))


def test_keep_threshold_and_scale():
    assert dr.keep16_of(0.1) == 58982 and dr.keep16_of(0.5) == 32768 and dr.keep16_of(0.0) == 65536
    assert float(dr.inv_keep_of(0.1)) == pytest.approx(1 / 0.9, rel=1e-5)  # nn.Dropout's 1 / (1 - p) to 6e-6
    assert float(dr.inv_keep_of(0.5)) == 2.0 and float(dr.inv_keep_of(0.0)) == 1.0


@pytest.mark.parametrize("variant,hidden", [(dr.GROUP, 64), (dr.TILE, 64), (dr.TILE, 128), (dr.QUAD, 32), (dr.QUAD, 64),
                                            (dr.QUAD, 128), (dr.Q16, 64), (dr.Q16, 128)])
def test_every_unit_gets_its_own_draw(variant, hidden):
    """The 16-bit draws of one row: hidden * layers values from distinct (block, stream, advance, half) positions -- no
    two units of a layer share a draw (a shared draw shows up as equal 16-bit values far beyond the birthday rate)."""
    bits = dr.train_bits(variant, hidden, 3, np.arange(4000), 7, 1234567)
    assert bits.shape == (3, 4000, hidden) and bits.max() < 65536
    flat = bits.transpose(1, 0, 2).reshape(4000, -1).astype(np.int64)
    n = flat.shape[1]
    srt = np.sort(flat, axis=1)
    dup = (srt[:, 1:] == srt[:, :-1]).sum(axis=1).mean()
    assert dup < 3 * n * (n - 1) / 2 / 65536 + 0.05  # birthday expectation n (n - 1) / 2 / 65536 per row
    # uniform 16-bit values: mean 32767.5, sd 18918
    assert abs(flat.mean() - 32767.5) < 5 * 18918 / np.sqrt(flat.size)


def test_masks_are_bernoulli_and_independent():
    rows = np.arange(20_000)
    for p in (0.1, 0.5):
        q = dr.keep16_of(p) / 65536.0
        a = dr.train_masks(dr.QUAD, 128, 3, rows, 1, 42, p)
        sd = np.sqrt(q * (1 - q) / a[0].size)
        assert all(abs(a[j].mean() - q) < 5 * sd for j in range(3))
        assert np.abs(a.mean(axis=(0, 1)) - q).max() < 6 * np.sqrt(q * (1 - q) / (3 * rows.size))  # per unit
        agree = q * q + (1 - q) * (1 - q)
        others = [dr.train_masks(dr.QUAD, 128, 3, rows, 2, 42, p), dr.train_masks(dr.QUAD, 128, 3, rows, 1, 43, p),
                  dr.train_masks(dr.TILE, 128, 3, rows, 1, 42, p), dr.train_masks(dr.Q16, 128, 3, rows, 1, 42, p), a[[1, 2, 0]], a[:, ::-1], a[:, :, ::-1],
                  np.roll(a, 1, axis=2), np.roll(a, 4, axis=2), np.roll(a, 32, axis=2), np.roll(a, 1, axis=1)]
        for b in others:
            assert abs((a == b).mean() - agree) < 6 * np.sqrt(agree * (1 - agree) / a.size)
    ap = dr.apply_masks(64, 2, rows, 100, 9, 0.1)
    ap2 = dr.apply_masks(64, 2, rows, 101, 9, 0.1)
    tr = dr.train_masks(dr.GROUP, 64, 2, rows, 100, 9, 0.1)
    q = dr.keep16_of(0.1) / 65536.0
    agree = q * q + (1 - q) ** 2
    for b in (ap2, tr):
        assert abs((ap == b).mean() - agree) < 6 * np.sqrt(agree * (1 - agree) / ap.size)
    assert dr.train_masks(dr.GROUP, 64, 2, rows[:10], 1, 1, 0.0).all()


def test_masked_forward_is_torch_dropout_arithmetic():
    """nn.Dropout(p) in training mode multiplies by mask / (1 - p) (options_model_3.py:85-103): mlp_forward_masked with a
    mask is the torch module's forward pass with its Bernoulli draw replaced by that mask."""
    torch = pytest.importorskip("torch")
    torch.manual_seed(0)
    x = torch.randn(4096, 128)
    y = torch.nn.functional.dropout(x, 0.1, training=True)
    kept = y != 0
    assert torch.allclose(y[kept] / x[kept], torch.full_like(y[kept], 1 / 0.9), rtol=1e-6)
    assert abs(float(kept.float().mean()) - 0.9) < 0.005
    H, L, n = 64, 2, 300
    lin = [torch.nn.Linear(7, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    state = {}
    for i, l_ in zip((0, 3, 6), lin):
        state[f"net.{i}.weight"] = l_.weight.detach().numpy()
        state[f"net.{i}.bias"] = l_.bias.detach().numpy()
    masks = dr.train_masks(dr.QUAD, H, L, np.arange(n), 3, 5, 0.1)
    xin = torch.randn(n, 7)
    h = xin
    for j in range(L):
        h = torch.relu(lin[j](h)) * torch.from_numpy(masks[j]).float() * float(dr.inv_keep_of(0.1))
    want = lin[2](h).detach().numpy()
    got = dr.mlp_forward_masked(state, xin.numpy(), masks, 0.1)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert np.allclose(dr.mlp_forward_masked(state, xin.numpy(), np.ones_like(masks), 0.0), rf.mlp_forward(state, xin.numpy()))


def test_oracle_reproduces_what_dropout_at_inference_does_to_the_references_price(golden):
    """SURVEY F5 on the reference's own trained 3 x 128 net and its paths (the fixture holds 1,024 paths x 50 steps): eval mode 7.2142 (reproduced exactly
    by the oracle), the reference's own run with nn.Dropout active 7.0225 (torch's mask stream).  Under the build's
    masks the oracle gives the same drop: noise on the continuation value, under the sticky rule, only ever triggers
    EARLIER exercise.  Measured over seeds 1..5: 6.958 .. 7.055 (mean 6.997)."""
    nn = golden["nn"]
    tag = "gbm_put"
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    S = nn[f"{tag}_S"]
    N = S.shape[0] - 1
    state = {k[len(tag) + 4:]: nn[k] for k in nn.files if k.startswith(f"{tag}_sd_")}
    fm, fs = nn[f"{tag}_feat_mean"], nn[f"{tag}_feat_std"]
    ym, ysd = nn[f"{tag}_Y_mean_std"]
    ev = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm, fs, ym, ysd)
    cf, ex, _ = rf.lsm_two_pass(S, K, r, T, True, *ev)
    p_eval, p_ref = float(nn[f"{tag}_price_eval"]), float(nn[f"{tag}_price_ref"])
    assert float(cf.mean()) == pytest.approx(p_eval, rel=1e-12) and np.array_equal(ex, nn[f"{tag}_ex_eval"])
    prices, fracs, drops = [], [], []
    for seed in (1, 2, 3):
        on = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm, fs, ym, ysd,
                                              dropout=dict(p=0.1, seed=seed, hidden=int(hidden), layers=3))
        cf_on, ex_on, _ = rf.lsm_two_pass(S, K, r, T, True, *on)
        prices.append(float(cf_on.mean()))
        d = cf_on - cf  # the same 1,024 paths with and without the masks: a paired comparison
        drops.append((float(d.mean()), float(d.std()) / np.sqrt(d.size)))
        fracs.append(float(ex_on.mean()))
        again, _, _ = rf.lsm_two_pass(S, K, r, T, True, *on)
        assert np.array_equal(again, cf_on)  # counter-based: the same call draws the same masks
    d_ref = nn[f"{tag}_cf_ref"] - cf                                 # the reference's own dropout-on run, same paths
    se_ref = float(d_ref.std()) / np.sqrt(d_ref.size)                # -0.192 +- 0.084
    assert all(dm < -1.5 * se for dm, se in drops)                   # the drop is there for every mask seed (-0.25, -0.20, -0.16) ...
    assert abs(np.mean([dm for dm, _ in drops]) - float(d_ref.mean())) < se_ref  # ... and is the reference's
    assert abs(np.mean(prices) - p_ref) < 0.1                        # 7.2142 -> 7.0225 there, -> 7.01 here
    assert all(f > float(ex.mean()) for f in fracs)                  # more paths exercise (earlier), as in ex_ref
    assert float(nn[f"{tag}_ex_ref"].mean()) > float(ex.mean())
    # p -> 0 is eval mode
    off = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm, fs, ym, ysd,
                                           dropout=dict(p=0.0, seed=1, hidden=int(hidden), layers=3))
    cf0, ex0, _ = rf.lsm_two_pass(S, K, r, T, True, *off)
    assert np.array_equal(ex0, ex) and np.allclose(cf0, cf, rtol=0, atol=0)
