"""look-up decoders (HBM-bound gather-add) -- bit-exact (fp32 adds in the reference's order) against
tests/golden/lut_decoders.npz, i.e. against what the REFERENCE's own PairwiseDecoderIVF.forward / map_codes and
reconstruct_from_fixed_codebooks returned (make_golden.py run_lut_case), and against the oracle restatement on fresh inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reconstruct_from_fixed_codebooks_bit_exact():
    from oracle.qinco_oracle import reconstruct_from_fixed_codebooks as ref
    from qinco_amd.lut import reconstruct_from_fixed_codebooks
    rs = np.random.RandomState(0)
    for (n, M, K, D, dt) in [(1000, 8, 256, 128, np.uint8), (257, 16, 256, 96, np.int64), (1, 4, 64, 32, np.int32),
                             (5000, 9, 256, 768, np.int64)]:
        cb = rs.randn(M, K, D).astype(np.float32)
        codes = rs.randint(0, K, (n, M)).astype(dt)
        got = reconstruct_from_fixed_codebooks(codes, cb)
        assert got.dtype == np.float32 and np.array_equal(got, ref(codes.astype(np.int64), cb))
    assert reconstruct_from_fixed_codebooks(np.zeros((0, 4), np.int64), rs.randn(4, 8, 16).astype(np.float32)).shape == (0, 16)
    with pytest.raises(IndexError):
        reconstruct_from_fixed_codebooks(np.full((3, 4), 8, np.int64), rs.randn(4, 8, 16).astype(np.float32))


def test_pairwise_decoder_bit_exact_and_device_path():
    import torch
    from oracle.qinco_oracle import pairwise_decoder_forward as ref
    from qinco_amd.lut import PairwiseDecoder
    rs = np.random.RandomState(1)
    M, K, D, IVF_M, ivf_K, Mt, n = 8, 32, 128, 5, 200, 11, 3000
    cb = rs.randn(Mt, K * K, D).astype(np.float32)
    comb = rs.randint(0, M + IVF_M, (2, Mt))
    imap = rs.randint(0, K, (ivf_K, IVF_M))
    codes_MB = rs.randint(0, K, (M, n))
    ivf = rs.randint(0, ivf_K, n)
    dec = PairwiseDecoder(cb, comb, K, imap)
    want = ref(codes_MB, ivf, cb, comb, K, imap)
    got = dec(codes_MB, ivf)
    assert np.array_equal(got, want)
    cols = torch.from_numpy(dec.gather_codes(codes_MB, ivf)).cuda()
    assert np.array_equal(dec._dec(cols).cpu().numpy(), want)
    dec.close()


def test_lookup_decoders_equal_the_reference_fixture():
    """The pin of row f4: lut_decode_kernel through qinco_amd.lut against the reference's recorded outputs -- same bits, host path and
    device path, int64 and int32 code columns (the search hands over int32, search_tasks.py:427-445)."""
    import torch
    from conftest import load_golden, lut_fixture_cases
    from qinco_amd.lut import PairwiseDecoder, reconstruct_from_fixed_codebooks
    for name, cb, comb, K, imap, codes_MB, ivf, mapped, xhat in lut_fixture_cases():
        dec = PairwiseDecoder(cb, comb, K, imap)
        cols = dec.gather_codes(codes_MB, ivf)
        assert np.array_equal(cols[:, comb[0]] * K + cols[:, comb[1]], mapped.T), name     # the columns map_codes combines
        got = dec(codes_MB, ivf)
        assert got.dtype == np.float32 and np.array_equal(got, xhat), name
        assert np.array_equal(dec(codes_MB.astype(np.int32), ivf.astype(np.int32)), xhat), name
        assert np.array_equal(dec._dec(torch.from_numpy(cols.astype(np.int32)).cuda()).cpu().numpy(), xhat), name
        dec.close()
    g = load_golden("lut_decoders")
    for name in ("u8", "i64", "one"):
        got = reconstruct_from_fixed_codebooks(g[f"fixed_{name}_codes"], g[f"fixed_{name}_codebooks"])
        assert np.array_equal(got, g[f"fixed_{name}_recons"]), name
