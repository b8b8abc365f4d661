"""The scenario generator's oracle: Philox4x32-10 against the published
known-answer vectors, exactness of the bit-to-double step, and the shape of
what it generates."""
import numpy as np

from oracle import scene_oracle as G

# Random123 kat_vectors, philox4x32 10 rounds: counter[4], key[2] -> output[4]
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = G.philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], key)
        assert tuple(int(x[0]) for x in got) == want


def test_uniform_is_exact_and_in_range():
    env = np.arange(10000)
    u0, u1 = G.uniform2(12345678901234, env, env * 0 + 3, G.K_CIRCLE_POS, env % 7)
    for u in (u0, u1):
        assert (u >= 0).all() and (u < 1).all()
        assert np.array_equal(u * 9007199254740992.0, np.floor(u * 9007199254740992.0))   # 53-bit integers
        assert abs(u.mean() - 0.5) < 0.02
    assert not np.array_equal(u0, u1)


def test_scene_is_a_function_of_env_and_episode_only():
    a = G.generate(7, np.arange(50), np.zeros(50, int), 4, 3, 5)
    b = G.generate(7, np.arange(50)[::-1].copy(), np.zeros(50, int), 4, 3, 5)
    for k in a:
        ax = 1 if k == "init_state" else 0
        assert np.array_equal(a[k], np.flip(b[k], axis=ax)), k
    c = G.generate(7, np.arange(50), np.ones(50, int), 4, 3, 5)
    assert not np.array_equal(a["circles"], c["circles"])
    # nothing is generated on top of the vessel; ships are well-formed
    d = np.hypot(a["circles"][..., 0], a["circles"][..., 1]) - a["circles"][..., 2]
    assert (d >= 25.0 - 1e-9).all()
    assert (a["poly_nv"] == 5).all()
    assert np.allclose(a["poly_pose"][..., 2] ** 2 + a["poly_pose"][..., 3] ** 2, 1.0)
