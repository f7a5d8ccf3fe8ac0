"""-m gpu: the pipelined real-bitstream GOP codec (vcamd/bitstream.py) against the CLI functions it accelerates.

encode_B.py:71-126 / decode_B.py:63-104 define the format and the arithmetic; the stream codec may only change WHEN things
happen (one analysis pass, asynchronous copies, coder threads, level batching), never a byte or a pixel."""
import numpy as np
import pytest
import torch

from helpers import lhbdc_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    _, prod = lhbdc_pair(1234, dev)
    prod.mv_compressor.update(force=True)
    prod.residual_compressor.update(force=True)
    return prod


def _gop(dev, seed, h=192, w=256):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, h + 8, w + 24, generator=g), 9, 1)
    return [base[..., :h, 2 * i:2 * i + w].contiguous().to(dev) for i in range(9)]


def test_stream_codec_frames_equal_the_cli_functions(dev, model):
    """Two frames batched through encode_frames: containers byte-identical to encode_B + write_container of each frame alone,
    reconstruction == decode_B of that container == decode_frames, bit for bit."""
    from vcamd import bitstream, lhbdc
    frames = _gop(dev, 3)
    codec = bitstream.LhbdcStreamCodec(model, workers=3)
    xb = torch.cat([frames[0], frames[2]], 0)
    xc = torch.cat([frames[1], frames[3]], 0)
    xa = torch.cat([frames[2], frames[4]], 0)
    with torch.no_grad():
        x_hat, pending = codec.encode_frames(xb, xc, xa)
        containers = pending.result()
        dec = codec.decode_frames(xb, xa, containers)
        assert torch.equal(dec, x_hat)
        for i in range(2):
            mv_bits, res_bits = lhbdc.encode_B(model, xa[i:i + 1], xc[i:i + 1], xb[i:i + 1])
            blob = lhbdc.write_container(None, 1626, mv_bits, res_bits)
            assert blob == containers[i], i
            _, s_mv, s_res, sh_mv, sh_res = lhbdc.read_container(containers[i])
            one = lhbdc.decode_B(xb[i:i + 1], xa[i:i + 1], model, s_mv, s_res, sh_mv, sh_res)
            assert torch.equal(one, x_hat[i:i + 1]), i
    codec.close()


def test_stream_codec_gop_roundtrip(dev, model):
    """A whole GOP-8: the decoder, fed only the seven containers and the two boundary frames, reproduces the encoder's
    reconstructions exactly, in hierarchical order (every level decodes from the level above's DECODED frames); coding it
    twice gives identical bytes; a corrupted container is refused or yields a different frame, never a crash."""
    from vcamd import bitstream, hip
    frames = _gop(dev, 5)
    codec = bitstream.LhbdcStreamCodec(model, workers=4)
    with torch.no_grad():
        containers, recon = codec.encode_gop(frames, frames[0], frames[8])
        again, _ = codec.encode_gop(frames, frames[0], frames[8])
        assert containers == again and sorted(containers) == [1, 2, 3, 4, 5, 6, 7]
        decoded = codec.decode_gop(containers, frames[0], frames[8])
        for o in range(1, 8):
            assert torch.equal(decoded[o], recon[o]), o
        assert all(len(c) > 24 for c in containers.values())
        bad = dict(containers)
        blob = bytearray(bad[4])
        blob[40] ^= 0xFF
        bad[4] = bytes(blob)
        try:
            other = codec.decode_gop(bad, frames[0], frames[8])
            assert not torch.equal(other[4], recon[4])
        except hip.VcError:
            pass
    codec.close()


def test_stream_codec_rejects_mixed_shapes_and_counts(dev, model):
    from vcamd import bitstream, hip
    frames = _gop(dev, 7)
    codec = bitstream.LhbdcStreamCodec(model, workers=2)
    with torch.no_grad():
        _, pending = codec.encode_frames(frames[0], frames[1], frames[2])
        (c,) = pending.result()
        with pytest.raises(hip.VcError):
            codec.decode_frames(torch.cat([frames[0], frames[0]], 0), torch.cat([frames[2], frames[2]], 0), [c])
    codec.close()
