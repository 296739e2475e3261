"""Tile tables of the active-set 3^3 convolution with staged-neighbourhood reuse (csrc/conv_brick.hip): built once per step and row
set from the set's flag volume (geometry phase), consumed by the forward and the data-gradient launches of the head layers
(conerf/model/feature_pyramid_net.py:47-56,97-103)."""
import torch

from . import lib as L

HCAP, TROWS, TAPS = 1280, 256, 28


class BrickTiles:
    """Device tables of one row set: rows in brick-major order, tiles, staged-voxel lists, (row, tap) -> LDS offset tables."""
    __slots__ = ("meta", "rows_sorted", "tiles", "halo", "nbr", "ntiles", "nrows", "dims", "overflow", "_ws")

    def record_stream(self, st):
        for t in (self.meta, self.rows_sorted, self.tiles, self.halo, self.nbr):
            t.record_stream(st)


def max_tiles_for(max_rows: int, B: int) -> int:
    return 2 * ((max_rows + TROWS - 1) // TROWS) + B + 64


def build_async(flags: torch.Tensor, max_rows: int) -> BrickTiles:
    """flags uint8 [B,D,H,W].  Enqueues the builder; finish(bt) after the stream has been synchronised (or meta read back) fills the counts."""
    lib = L.load()
    B, D, H, W = flags.shape
    dev = flags.device
    bt = BrickTiles()
    bt.dims = (B, D, H, W)
    mt = max_tiles_for(max_rows, B)
    nb = int(lib.dreg_brick_tiles_workspace_bytes(B, D, H, W))
    bt._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    bt.meta = torch.zeros(4, dtype=torch.int32, device=dev)
    bt.rows_sorted = torch.empty(max(max_rows, 1), dtype=torch.int32, device=dev)
    bt.tiles = torch.empty(mt, 4, dtype=torch.int32, device=dev)
    bt.halo = torch.empty(mt, HCAP, dtype=torch.int32, device=dev)
    bt.nbr = torch.empty(mt, TROWS, TAPS, dtype=torch.int16, device=dev)
    L.check(lib.dreg_brick_tiles_build(L.ptr(flags), B, D, H, W, max_rows, mt, L.ptr(bt._ws), nb, L.ptr(bt.meta), L.ptr(bt.rows_sorted),
                                       L.ptr(bt.tiles), L.ptr(bt.halo), L.ptr(bt.nbr), L.stream()), "dreg_brick_tiles_build")
    bt.ntiles = bt.nrows = -1
    bt.overflow = False
    return bt


def finish(bt: BrickTiles, meta_host) -> BrickTiles:
    bt.nrows, _, bt.ntiles, ov = (int(v) for v in meta_host)
    bt.overflow = bool(ov) or bt.ntiles > bt.tiles.shape[0]
    return bt


def build(flags: torch.Tensor, max_rows: int) -> BrickTiles:
    bt = build_async(flags, max_rows)
    return finish(bt, bt.meta.tolist())


def pack_weight(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """w fp32 [Cout,Cin,3,3,3] -> the bf16 operand pack of dreg_conv3_brick (forward, or the flipped-tap data-gradient form)."""
    lib = L.load()
    cout, cin = w.shape[0], w.shape[1]
    rows, red = (cin, cout) if transposed else (cout, cin)
    out = torch.empty(lib.dreg_conv3_brick_pack_bytes(rows, red) // 2, dtype=torch.bfloat16, device=w.device)
    L.check(lib.dreg_pack_conv_weight_brick(L.ptr(w.detach().contiguous()), L.ptr(out), cout, cin, int(transposed), L.stream()), "dreg_pack_conv_weight_brick")
    return out


def conv(x, wpk, out, bias, addend, bt: BrickTiles, cin: int, cout: int, add_same: bool = False):
    """out[rows of bt] = bias + addend + conv3(x); x [B,D,H,W,cin] bf16, out [B,D,H,W,cout] (bf16 or fp32), other rows untouched."""
    lib = L.load()
    B, D, H, W = bt.dims
    Da = Ha = Wa = 0
    if addend is not None:
        Da, Ha, Wa = addend.shape[1:4]
    L.check(lib.dreg_conv3_brick(L.ptr(x), L.ptr(wpk), L.ptr(out), L.ptr(bias), L.ptr(addend), L.ptr(bt.tiles), bt.ntiles, L.ptr(bt.halo), L.ptr(bt.nbr),
                                 L.ptr(bt.rows_sorted), B, D, H, W, cin, cout, Da, Ha, Wa, int(add_same), int(out.dtype == torch.float32), L.stream()),
            "dreg_conv3_brick")
    return out
