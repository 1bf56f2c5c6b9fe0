"""ISA lint of the ticket-scheduled corner kernel (csrc/corner.hip, corner_stream_kernel) -- CPU only.

The kernel's correctness rests on hand-counted `s_waitcnt vmcnt(N)` waits (vmcnt retires in issue order), on an
inline-asm returning atomic, and on the compiler emitting NO other vector-memory operation between the ticket draw
and the counted wait.  Round 2's GPU memory fault came from exactly this region; the assumptions were checked by hand
in the .s at the time.  This test checks them on every build: the product object's gfx950 assembly is kept by
`-save-temps=obj` (build.py) and every instance of the kernel must show the structure the waits were counted for.  A
compiler bump that reorders, adds or drops a VMEM operation there fails HERE, on the CPU, not as a fault on the GPU.
Reference semantics of the kernel: pytorch_model/detector/shi_tomasi.py:66-112."""
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ASM = os.path.join(ROOT, "onnx_image_processing_amd", "lib", "obj", "corner-hip-amdgcn-amd-amdhsa-gfx950.s")
TW, LPAD = 128, 4                                   # csrc/corner.hip: tile width, LDS padding per side
INSTANCES = [(3, 4, 0), (3, 5, 0), (3, 8, 0), (3, 4, 1), (3, 5, 1), (3, 8, 1), (5, 4, 0), (5, 5, 0), (5, 8, 0), (7, 8, 0)]     # (block, rows per thread, uint8 pixels)


@pytest.fixture(scope="module")
def kernels():
    from onnx_image_processing_amd.build import build
    build(verbose=False)
    if not os.path.exists(ASM):                     # an object built before -save-temps was added
        build(force=True, verbose=False)
    text = open(ASM).read()
    out = {}
    pat = (r"^(_ZN12_GLOBAL__N_120corner_stream_kernelILi(\d+)ELi(\d+)ELb([01])EEEv6MiSetsPfiiiiiPjPy):[^\n]*\n(.*?)"
           r"^\.Lfunc_end\d+:")
    for m in re.finditer(pat, text, flags=re.S | re.M):
        body = [ln.strip() for ln in m.group(5).split("\n") if ln.startswith("\t")]
        ins = [ln for ln in body if not ln.startswith((".", ";"))]
        desc = text[text.index(".amdhsa_kernel " + m.group(1)):]
        desc = desc[:desc.index(".end_amdhsa_kernel")]
        out[(int(m.group(2)), int(m.group(3)), int(m.group(4)))] = (ins, desc)
    return out


def _expect(bs, r, u8):
    hl = bs // 2 + 1
    lh = 8 * r + 2 * hl
    lw4 = (TW + 2 * LPAD) // 4
    nch = (lh * lw4 + 255) // 256                   # DMA pieces per wave per tile
    return nch, 2 * nch * 256 * (4 if u8 else 16)   # and the two tile buffers, nothing else, in LDS


def test_every_instance_is_in_the_object(kernels):
    assert sorted(kernels) == sorted(INSTANCES)


@pytest.mark.parametrize("key", INSTANCES)
def test_ticket_region_has_exactly_the_counted_operations(kernels, key):
    bs, r, u8 = key
    ins, desc = kernels[key]
    nch, lds_bytes = _expect(bs, r, u8)
    draw = [i for i, ln in enumerate(ins) if "MI_TICKET_DRAW" in ln]
    wait = [i for i, ln in enumerate(ins) if "MI_TICKET_WAIT" in ln]
    assert len(draw) == 1 and len(wait) == 1 and draw[0] < wait[0]
    # the draw is the inline-asm returning atomic, data and destination in AGPRs (no compiler-visible result that
    # would make hipcc wait vmcnt(0) where the value joins control flow)
    assert re.match(r"global_atomic_add a\d+, v\d+, a\d+, s\[\d+:\d+\] sc0", ins[draw[0]]), ins[draw[0]]
    assert ins[wait[0]].startswith(f"s_waitcnt vmcnt({nch + r})"), ins[wait[0]]

    def vmcnt(ln):
        m = re.match(r"s_waitcnt\b.*vmcnt\((\d+)\)", ln)
        return int(m.group(1)) if m else None

    waits = [(i, vmcnt(ln)) for i, ln in enumerate(ins) if vmcnt(ln) is not None]
    # the whole kernel has five vmcnt waits: [vmcnt(0) | vmcnt(R)] at the top of a tile (previous tile partial / full),
    # [vmcnt(0)] + [vmcnt(NCH + R)] at the ticket, and one vmcnt(0) behind the final done-counter atomic
    assert [v for _, v in waits] == [0, r, 0, nch + r, 0], waits
    top0, topr, pre, tick, last = [i for i, _ in waits]
    assert topr < draw[0] < pre == tick - 1 and tick == wait[0] < last
    # between the draw and the counted wait, in program order: exactly the NCH DMA pieces of the next tile and the R
    # stores of this tile -- no other vector-memory instruction (each would shift the count), no scratch (spills are
    # VMEM operations too)
    region = ins[draw[0] + 1:wait[0]]
    vmem = [ln for ln in region if ln.split()[0].startswith(("global_", "buffer_", "flat_", "scratch_"))]
    dma = [ln for ln in vmem if ln.startswith("global_load_lds_dword" + ("x4" if not u8 else ""))]
    if u8:
        dma = [ln for ln in dma if not ln.startswith("global_load_lds_dwordx")]
    stores = [ln for ln in vmem if ln.startswith("global_store_dwordx4")]
    assert len(dma) == nch and len(stores) == r and len(vmem) == nch + r, vmem
    # all DMA pieces precede all stores (the wait at the top of the next tile, vmcnt(R), relies on it)
    assert max(region.index(x) for x in dma) < min(region.index(x) for x in stores)
    # no vmcnt wait sits between the DMA issue and the tile's first LDS read other than the counted ones above
    assert not [v for i, v in waits if draw[0] < i < pre]
    # one barrier per tile, ahead of the draw (the wait blocks in front of it are laid out after it in the text)
    barriers = [i for i, ln in enumerate(ins) if ln.startswith("s_barrier")]
    assert len(barriers) == 1 and barriers[0] < draw[0]
    # the prologue issues the first tile's NCH pieces: 2 * NCH DMA instructions in the kernel
    assert len([ln for ln in ins if ln.startswith("global_load_lds_")]) == 2 * nch
    # whole kernel: no scratch, and LDS = the two tile buffers exactly (fp32: 2 x 20 KiB = four workgroups per CU;
    # the ticket rides in a scratch slot of a buffer, not in an extra word)
    assert not [ln for ln in ins if ln.startswith("scratch_")]
    assert re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", desc).group(1) == str(lds_bytes)
    assert re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1) == "0"
