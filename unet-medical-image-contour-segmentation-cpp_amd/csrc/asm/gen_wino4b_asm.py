#!/usr/bin/env python3
"""gen_wino4b_asm.py -- emits the gfx950 assembly of conv3x3_wino4b_f32: Winograd F(4x4,3x3) for the layers with 64 output channels
per workgroup (the fp32 plan's 512^2-resolution layers), persistent, hand-scheduled.

Why a second shape.  conv3x3_wino4a_f32 gives a wave 16 tiles x 32 channels: at Cout = 64 that is half a workgroup.
conv3x3_wino4s_f32 (hipcc) gives a wave 16 tiles x 16 channels, 144 accumulators, two workgroups per CU: twice the U and V operand
traffic per MFMA, and a prologue / epilogue per 16 tiles x 64 channels.  Here a wave owns 32 tiles x 16 channels -- the same 288
accumulators as the two-block kernel, with the two accumulator blocks being two TILE GROUPS (the left and the right 16x16 pixels of
a 16x32 block) that share every U fragment: half the U bytes per MFMA of either other kernel.

  * workgroup = 4 waves = a block of 16 x 32 output pixels (32 tiles) x 64 channels, one per CU, persistent over its XCD's blocks;
  * V = B^T d B of a 16-channel chunk for 32 tiles is 72 KB: ONE buffer, so a chunk is two phases -- transform (raw patch -> V), barrier,
    288 MFMAs per wave -- which costs nothing against threading the transform between the MFMAs: on gfx950 the fp32 MFMA never
    co-executes with VALU work (profiles/r04_fp32_mfma_filler_probe.txt), only the LDS latencies of the transform are exposed;
  * the raw patch (18 x 34 pixels x 16 channels) is double-buffered and arrives by LDS-DMA two chunks ahead, during the MFMA phase;
  * U ring, exact wait counts, tile-to-tile hand-over (next block's first two raw chunks and its U ring requested inside the last two
    chunks, every load home before the epilogue's stores), in-lane inverse transform: as in gen_wino4_asm.py.

Shape contract (csrc/wino4_asm.cpp): H % 16 == 0, W % 32 == 0, Cin % 32 == 0 and >= 64, Cout % 64 == 0, fp32, no fused head.
usage: gen_wino4b_asm.py out.s [--stop N [--dump lds|vgpr|agpr]]
"""
import sys

from gen_wino4_asm import Emitter, LdsQueue, a, s, v, waitcnt

UD = 12                      # U ring depth in positions (one 16-channel fragment = 4 registers per position; same-card A/B of 6 / 9 / 12:
                             # up4.c1 1.543 / 1.518 / 1.506 ms, profiles/r04_ab_wino4b_variants.txt)
VPOS_B = 2048                # bytes per position of V: 32 tiles x 64 B
VBUF_B = 36 * VPOS_B         # 73728
ROWP = 36                    # pixel slots per patch row (34 live): pixel x = 4 a + c sits in slot 9 c + a
RAW_SLOTS = 18 * ROWP        # 648
RAW_LOADS = 44               # wave-wide LDS-DMA loads per chunk (41 live; eleven per wave)
DMA_PER_WAVE = 11
RAWBUF_B = RAW_LOADS * 1024  # 45056
LDS_R0 = VBUF_B
LDS_BYTES = LDS_R0 + 2 * RAWBUF_B    # 163840 = the whole LDS of a CU
STOP_AT = 0
DUMP = ""

# ---------------------------------------------------------------------------------------------------------------- registers
L_RIDX, L_M0CUR = 0, 1       # (unused here: s0 / s1 are free once the arguments are loaded)
S_KARG, S_WG = 0, 2
S_T0, S_T1, S_T2, S_T3, S_T4 = 3, 4, 5, 6, 7
A_IN, A_U, A_BIAS, A_OUT, A_POOL = 8, 10, 12, 14, 16
A_H, A_W, A_PIXIN, A_NCH, A_TX, A_TY, A_MT, A_NWG = 18, 19, 20, 21, 22, 23, 24, 25
A_MAGM, A_MAGX, A_MAGY, A_UPOS, A_UBYTES, A_IMGIN = 26, 27, 28, 29, 30, 31
A_PIXOUT, A_COOFF, A_IMGOUT, A_PIXPOOL, A_IMGPOOL, A_RELU, A_GRID, A_FLAGS = 32, 33, 34, 35, 36, 37, 38, 39
R_IN, R_U, R_OUT, R_POOL = 40, 44, 48, 52
N_OUTB, N_POOLB = 56, 58
N_TOUT, N_TPOOL, N_UBASE, N_HAS = 60, 61, 62, 63
L_TT, L_TCOUNT, L_TSTART, L_SLOTS = 64, 65, 66, 67
L_UOFF, L_DMAOFF, L_PAIRS, L_WAVE = 68, 69, 70, 71
K_4, K_M5, K_2, K_M2, K_ALPHA, K_BETA, K_MBETA, K_M4, K_8 = 72, 74, 76, 78, 80, 82, 84, 86, 88
C_TOUT, C_TPOOL = 90, 91
L_M0BASE = 92
L_BY0M1, L_BX0M1, L_BASEPIX = 93, 94, 95
S_ROW = 96                   # s[96:99]: per-row scalar offsets of the epilogue
S_G1, S_PG1 = 100, 101       # byte offset of the right tile group in an output row / a pooled row
S_PROW = 0                   # s[0:1]: pooled rows (the argument pointer is dead by then)

U0 = 0
ACCV = 224


def layout(ud):
    """vector register map for a U ring of `ud` positions (4 registers each)"""
    global UD, AV0, PX0, CR0, E0, TV0, V_VRD0, V_VRD1, V_UVOFF, V_PRD, V_VWR, V_PRD2, V_VWR2, V_REL0, V_VOFF0, V_PYPX0
    global V_OUTOFF, V_POOLOFF, V_BIAS0, V_CHAN4, V_T0, EP_M, EP_S, EP_Y, EP_C, EP_T, EP_VX, EP_VXP
    UD = ud
    AV0 = 4 * UD                 # [2 position parities][2 groups][4]
    PX0 = AV0 + 16
    CR0 = PX0 + 32
    E0 = CR0 + 48
    TV0 = E0 + 16
    base = TV0 + 12
    V_VRD0, V_VRD1, V_UVOFF, V_PRD, V_VWR, V_PRD2, V_VWR2 = (base + i for i in range(7))   # V_PRD / V_VWR: two-row pass; ..2: one-row pass
    V_REL0 = base + 7            # 11
    V_VOFF0 = V_REL0 + 11        # 11
    V_PYPX0 = V_VOFF0 + 11       # 11
    V_OUTOFF, V_POOLOFF, V_BIAS0, V_CHAN4 = (V_PYPX0 + 11 + i for i in range(4))
    V_T0 = V_CHAN4 + 1           # eight temporaries (set-up code only)
    assert V_T0 + 8 <= ACCV, (ud, V_T0)
    EP_M, EP_S, EP_Y, EP_C = PX0, PX0 + 12, PX0 + 20, PX0 + 28
    EP_T = CR0
    EP_VX = E0                   # 16 per-x voffsets
    EP_VXP = TV0                 # 8 per-x pooled voffsets


layout(UD)


class E2(Emitter):
    def checkpoint(self, n, what):
        self.c(f"checkpoint {n}: {what}")
        if STOP_AT == n:
            if DUMP:
                emit_dump(self, DUMP)
            self.i("s_branch .Lend_program")


def emit_dump(E, what):
    E.i(f"s_cmp_lg_u32 {s(S_WG)}, 0")
    E.i("s_cbranch_scc1 .Lend_program")
    E.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E.i("s_barrier")
    E.i(f"s_mov_b32 {s(S_ROW)}, {s(A_OUT)}")
    E.i(f"s_and_b32 {s(S_ROW + 1)}, {s(A_OUT + 1)}, 0xffff")
    E.i(f"s_mov_b32 {s(S_ROW + 2)}, 0x7ffffff0")
    E.i(f"s_mov_b32 {s(S_ROW + 3)}, 0x00020000")
    t, o = V_T0, V_T0 + 1
    E.i(f"v_mbcnt_lo_u32_b32 {v(t)}, -1, 0")
    E.i(f"v_mbcnt_hi_u32_b32 {v(t)}, -1, {v(t)}")
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(L_WAVE)}, 6")
    E.i(f"v_add_u32 {v(t)}, {s(S_T0)}, {v(t)}")
    if what == "lds":
        E.i(f"v_lshlrev_b32 {v(o)}, 4, {v(t)}")
        E.i(f"v_add_u32 {v(t)}, 61440, {v(o)}")
        E.i(f"v_add_u32 {v(V_T0 + 3)}, 122880, {v(o)}")
        for i in range(LDS_BYTES // 4096):
            base, off = (o, i * 4096) if i < 15 else (t, (i - 15) * 4096) if i < 30 else (V_T0 + 3, (i - 30) * 4096)
            E.i(f"ds_read_b128 {v(PX0, 4)}, {v(base)} offset:{off}")
            E.i("s_waitcnt lgkmcnt(0)")
            E.i(f"s_mov_b32 {s(S_T1)}, {i * 4096}")
            E.i(f"buffer_store_dwordx4 {v(PX0, 4)}, {v(o)}, {s(S_ROW, 4)}, {s(S_T1)} offen")
            E.i("s_waitcnt vmcnt(0)")
    else:
        E.i(f"v_lshlrev_b32 {v(o)}, 2, {v(t)}")
        for r in range(256):
            if r in (V_T0, V_T0 + 1, V_T0 + 2):
                continue
            E.i(f"s_mov_b32 {s(S_T1)}, {r * 1024}")
            if what == "agpr":
                E.i(f"v_accvgpr_read_b32 {v(V_T0 + 2)}, {a(r)}")
                E.i("s_nop 1")
                E.i(f"buffer_store_dword {v(V_T0 + 2)}, {v(o)}, {s(S_ROW, 4)}, {s(S_T1)} offen")
            else:
                E.i(f"buffer_store_dword {v(r)}, {v(o)}, {s(S_ROW, 4)}, {s(S_T1)} offen")
    E.i("s_waitcnt vmcnt(0)")


def acc_reg(p, g, r=0, cnt=4):
    """accumulators of position p, tile group g: AGPRs for p < 32, VGPRs for the last four positions"""
    if p < 32:
        return a(8 * p + 4 * g + r, cnt)
    return v(ACCV + 8 * (p - 32) + 4 * g + r, cnt)


def pk_fma(E, dst, k_sgpr, x, y):
    for h in (0, 2):
        E.i(f"v_pk_fma_f32 {v(dst + h, 2)}, {s(k_sgpr, 2)}, {v(x + h, 2)}, {v(y + h, 2)}")


def pk_add(E, dst, x, y):
    for h in (0, 2):
        E.i(f"v_pk_add_f32 {v(dst + h, 2)}, {v(x + h, 2)}, {v(y + h, 2)}")


def pk_sub(E, dst, x, y):
    for h in (0, 2):
        E.i(f"v_pk_add_f32 {v(dst + h, 2)}, {v(x + h, 2)}, {v(y + h, 2)} neg_lo:[0,1] neg_hi:[0,1]")


# -------------------------------------------------------------------------------------------------------------- transform
# One tile group at a time: the 16x16 pixels at column offset 16 g of the block; lane = (tile of the group, channel quad), the wave's
# role = rows of B^T as in gen_wino4_asm.py.  A whole chunk (both groups) runs with nothing beside it, so the only scheduling is the
# software pipeline of the patch reads: column k + 2 is requested while column k is combined.
def px_bank(k):
    return PX0 + 16 * (k & 1)


def cr(row, k):
    return CR0 + 4 * (6 * row + k)


def lq_reserve(E, lq, n):
    """lgkmcnt counts at most 15 outstanding LDS operations: before n more are issued, retire the oldest ones (long done)"""
    if len(lq.q) + n > 15:
        keep = min(len(lq.q), 15 - n - 3)
        E.i(f"s_waitcnt lgkmcnt({keep})")
        lq.q = lq.q[len(lq.q) - keep:]


def tr_load(E, role, g, k, rbuf, lq):
    """patch column k of this lane's rows (g only tags the pass: the tile group's columns are part of the base register)"""
    rows = 4 if role == "A" else 3
    rstep = 1 if role == "A" else 2
    prd = V_PRD if role == "A" else V_PRD2
    lq_reserve(E, lq, rows)
    for i in range(rows):
        off = (9 * (k & 3) + (k >> 2)) * 64 + i * rstep * ROWP * 64 + rbuf * RAWBUF_B
        assert off < 65536
        E.i(f"ds_read_b128 {v(px_bank(k) + 4 * i, 4)}, {v(prd)} offset:{off}")
        lq.issue(("L", g, k))


def tr_col(E, role, k):
    d = [px_bank(k) + 4 * i for i in range(4)]
    if role == "A":
        ta, tb = TV0, TV0 + 4
        pk_fma(E, ta, K_ALPHA, d[1], d[3])
        pk_fma(E, tb, K_ALPHA, d[0], d[2])
        pk_fma(E, cr(0, k), K_BETA, tb, ta)
        pk_fma(E, cr(1, k), K_MBETA, tb, ta)
    else:
        t = TV0
        pk_fma(E, t, K_4, d[0], d[2])
        pk_fma(E, cr(0, k), K_M5, d[1], t)


def tr_prep(E, row):
    c = [cr(row, k) for k in range(6)]
    pk_fma(E, E0, K_M4, c[2], c[4])
    pk_fma(E, E0 + 4, K_M4, c[1], c[3])
    pk_sub(E, E0 + 8, c[4], c[2])
    pk_sub(E, E0 + 12, c[3], c[1])


def tr_store_calc(E, row, nu, dst):
    c = [cr(row, k) for k in range(6)]
    if nu == 0:
        pk_fma(E, dst, K_4, c[0], c[4])
        pk_fma(E, dst, K_M5, c[2], dst)
    elif nu == 1:
        pk_add(E, dst, E0, E0 + 4)
    elif nu == 2:
        pk_sub(E, dst, E0, E0 + 4)
    elif nu == 3:
        pk_fma(E, dst, K_2, E0 + 12, E0 + 8)
    elif nu == 4:
        pk_fma(E, dst, K_M2, E0 + 12, E0 + 8)
    else:
        pk_fma(E, dst, K_4, c[1], c[5])
        pk_fma(E, dst, K_M5, c[3], dst)


def tr_write(E, role, g, row, nu, src, lq):
    off = nu * VPOS_B + row * 6 * VPOS_B
    assert off < 65536
    lq_reserve(E, lq, 1)
    E.i(f"ds_write_b128 {v(V_VWR if role == 'A' else V_VWR2)}, {v(src, 4)} offset:{off}")
    lq.issue(("W", g, row, nu))


def emit_transform(E, rbuf):
    """V = B^T d B of one chunk, Raw[rbuf] -> V, for both tile groups.  The work is cut into eight (tile group, rows of B^T) pieces --
    two rows (1,2) or (3,4): 96 packed operations, one row 0 or 5: 48 -- and every wave takes a two-row piece of one group and a
    one-row piece of the OTHER group: 144 packed operations each, no wave waits for a heavier one at the barrier.  Which pieces is
    in the wave's lane constants (V_PRD / V_VWR, V_PRD2 / V_VWR2, alpha / beta): one instruction stream for all four waves."""
    lq = LdsQueue()
    E.c(f"transform: Raw{rbuf} -> V")
    seq = [("A", 0, k) for k in range(6)] + [("B", 1, k) for k in range(6)]
    tr_load(E, seq[0][0], seq[0][1], seq[0][2], rbuf, lq)
    tr_load(E, seq[1][0], seq[1][1], seq[1][2], rbuf, lq)
    for n, (role, g, k) in enumerate(seq):
        waitcnt(E, lgkm=lq.wait_count(("L", g, k)))
        tr_col(E, role, k)
        if n + 2 < len(seq):
            tr_load(E, seq[n + 2][0], seq[n + 2][1], seq[n + 2][2], rbuf, lq)      # into the bank column k just left
        if k == 5:                                         # the pass's six columns are combined: its row pass
            for row in range(2 if role == "A" else 1):
                tr_prep(E, row)
                for grp in ((0, 1, 2), (3, 4, 5)):
                    for m, nu in enumerate(grp):
                        tr_store_calc(E, row, nu, TV0 + 4 * m)
                    for m, nu in enumerate(grp):
                        tr_write(E, role, g, row, nu, TV0 + 4 * m, lq)
                    E.i("s_nop 1")


# ------------------------------------------------------------------------------------------------------------- chunk body
ST_POLICY = ""               # cache policy of the output stores: the default one -- "nt", which pays in gen_wino4_asm.py, costs here (same card: up4.c1
                             # 1.45 -> 1.51 ms, profiles/r04_ab_store_policy.txt); tuning: --store-policy
DMA_POS = tuple(range(1, 34, 3))          # one LDS-DMA load every third position of the MFMA phase (tuning: --dma-pos; same card: the first
                                          # eleven positions 1.518 ms, every second 1.504, every third 1.47, positions 12..22 1.53)


def vm_wait_for_position(p):
    """U fragment of position p: issued at the end of position p - UD; younger: one refill per position since, and the LDS-DMA
    loads of positions p-UD+1 .. p-1 of this body (none at the tail of the previous one)"""
    dmas = sum(1 for q in DMA_POS if max(0, p - UD + 1) <= q < p)
    return (UD - 1) + dmas


def emit_body(E, par, kind):
    assert kind in ("first", "mid", "last")
    E.c(f"---- chunk body: Raw{par}, {kind}")
    # raw(c) is in LDS (requested two bodies ago: every U wait of the previous body implies it; the first tile waited for everything),
    # and every wave has finished the MFMA phase that read V: then the transform may overwrite V
    E.i("s_barrier")
    emit_transform(E, par)
    waitcnt(E, lgkm=0)
    E.i("s_barrier")
    lq = LdsQueue()
    for g in range(2):
        E.i(f"ds_read_b128 {v(AV0 + 4 * g, 4)}, {v(V_VRD0)} offset:{g * 1024}")
        lq.issue(("AV", 0, g))
    for p in range(36):
        avb = AV0 + 8 * (p & 1)
        slot = p % UD
        vm = None if (kind == "first" and p < UD) else vm_wait_for_position(p)
        waitcnt(E, vm=vm, lgkm=lq.wait_count(("AV", p, 1)))
        dma = p in DMA_POS
        for m in range(8):
            st, g = m >> 1, m & 1
            acc = acc_reg(p, g)
            csrc = "0" if (kind == "first" and p != 7 and st == 0) else acc
            if dma and m == 2:
                E.i(f"s_add_u32 m0, {s(L_M0BASE)}, {par * RAWBUF_B + 4096 * DMA_POS.index(p)}")
            E.i(f"v_mfma_f32_16x16x4_f32 {acc}, {v(avb + 4 * g + st)}, {v(U0 + 4 * slot + st)}, {csrc}")
            if m < 2 and p + 1 < 36:
                q = p + 1
                base, off = (V_VRD0, q * VPOS_B) if q < 32 else (V_VRD1, (q - 32) * VPOS_B)
                E.i(f"ds_read_b128 {v(AV0 + 8 * (q & 1) + 4 * m, 4)}, {v(base)} offset:{off + m * 1024}")
                lq.issue(("AV", q, m))
            if dma and m == 2:
                E.i(f"buffer_load_dwordx4 {v(V_VOFF0 + DMA_POS.index(p))}, {s(R_IN, 4)}, {s(L_DMAOFF)} offen lds")
            if m == 7:
                if kind == "last" and p == 36 - UD:
                    E.i(f"s_mov_b32 {s(L_UOFF)}, {s(N_UBASE)}")
                E.i(f"buffer_load_dwordx4 {v(U0 + 4 * slot, 4)}, {v(V_UVOFF)}, {s(R_U, 4)}, {s(L_UOFF)} offen")
        E.i(f"s_add_u32 {s(L_UOFF)}, {s(L_UOFF)}, {s(A_UPOS)}")
        if p == max(DMA_POS):
            E.i(f"s_add_u32 {s(L_DMAOFF)}, {s(L_DMAOFF)}, 64")


# ----------------------------------------------------------------------------------------------------------------- set-up
def mul64_add(E, dst_pair, base_pair, a_s, b_s):
    E.i(f"s_mul_i32 {s(S_T3)}, {s(a_s)}, {s(b_s)}")
    E.i(f"s_mul_hi_u32 {s(S_T4)}, {s(a_s)}, {s(b_s)}")
    E.i(f"s_add_u32 {s(dst_pair)}, {s(base_pair)}, {s(S_T3)}")
    E.i(f"s_addc_u32 {s(dst_pair + 1)}, {s(base_pair + 1)}, {s(S_T4)}")


def udiv_magic(E, q, n, magic):
    E.i(f"s_mul_hi_u32 {s(q)}, {s(n)}, {s(magic)}")
    E.i(f"s_cmp_eq_u32 {s(magic)}, 0")
    E.i(f"s_cselect_b32 {s(q)}, {s(n)}, {s(q)}")


def emit_setup_tile(E):
    """decode the logical block index in s[S_T0] and make it the NEXT block (gen_wino4_asm.py: emit_setup_tile); blocks are 16 rows x
    32 columns of pixels, channel groups of 64"""
    udiv_magic(E, S_T1, S_T0, A_MAGM)
    E.i(f"s_mul_i32 {s(S_T2)}, {s(S_T1)}, {s(A_MT)}")
    E.i(f"s_sub_u32 {s(S_T0)}, {s(S_T0)}, {s(S_T2)}")
    E.i(f"s_lshl_b32 {s(N_UBASE)}, {s(S_T1)}, 12")                           # n_tile * 64 channels * 64 bytes
    E.i(f"s_lshl_b32 {s(N_TOUT)}, {s(S_T1)}, 8")                             # n_tile * 64 channels * 4 bytes
    E.i(f"s_mov_b32 {s(N_TPOOL)}, {s(N_TOUT)}")
    E.i(f"v_add_u32 {v(V_T0)}, {s(N_TOUT)}, {v(V_CHAN4)}")
    E.i(f"global_load_dword {v(V_BIAS0)}, {v(V_T0)}, {s(A_BIAS, 2)}")
    udiv_magic(E, S_T1, S_T0, A_MAGX)
    E.i(f"s_mul_i32 {s(S_T2)}, {s(S_T1)}, {s(A_TX)}")
    E.i(f"s_sub_u32 {s(S_T2)}, {s(S_T0)}, {s(S_T2)}")                       # tx
    udiv_magic(E, S_T4, S_T1, A_MAGY)                                          # b
    E.i(f"s_mul_i32 {s(S_T3)}, {s(S_T4)}, {s(A_TY)}")
    E.i(f"s_sub_u32 {s(S_T3)}, {s(S_T1)}, {s(S_T3)}")                       # ty
    E.i(f"s_mov_b32 {s(S_T0)}, {s(S_T4)}")
    E.i(f"s_lshl_b32 {s(S_T1)}, {s(S_T3)}, 4")                               # by0
    E.i(f"s_lshl_b32 {s(S_T2)}, {s(S_T2)}, 5")                               # bx0 (32-pixel blocks)
    mul64_add(E, R_IN, A_IN, S_T0, A_IMGIN)
    E.i(f"s_and_b32 {s(R_IN + 1)}, {s(R_IN + 1)}, 0xffff")
    mul64_add(E, N_OUTB, A_OUT, S_T0, A_IMGOUT)
    E.i(f"s_and_b32 {s(N_OUTB + 1)}, {s(N_OUTB + 1)}, 0xffff")
    mul64_add(E, N_POOLB, A_POOL, S_T0, A_IMGPOOL)
    E.i(f"s_and_b32 {s(N_POOLB + 1)}, {s(N_POOLB + 1)}, 0xffff")
    E.i(f"s_mul_i32 {s(S_T3)}, {s(S_T1)}, {s(A_W)}")
    E.i(f"s_add_u32 {s(S_T3)}, {s(S_T3)}, {s(S_T2)}")
    E.i(f"s_mul_i32 {s(L_BASEPIX)}, {s(S_T3)}, {s(A_PIXIN)}")
    E.i(f"s_mul_i32 {s(S_T4)}, {s(S_T3)}, {s(A_PIXOUT)}")
    E.i(f"s_add_u32 {s(N_TOUT)}, {s(N_TOUT)}, {s(S_T4)}")
    E.i(f"s_lshr_b32 {s(S_T3)}, {s(S_T1)}, 1")
    E.i(f"s_lshr_b32 {s(S_T4)}, {s(A_W)}, 1")
    E.i(f"s_mul_i32 {s(S_T3)}, {s(S_T3)}, {s(S_T4)}")
    E.i(f"s_lshr_b32 {s(S_T4)}, {s(S_T2)}, 1")
    E.i(f"s_add_u32 {s(S_T3)}, {s(S_T3)}, {s(S_T4)}")
    E.i(f"s_mul_i32 {s(S_T3)}, {s(S_T3)}, {s(A_PIXPOOL)}")
    E.i(f"s_add_u32 {s(N_TPOOL)}, {s(N_TPOOL)}, {s(S_T3)}")
    E.i(f"s_sub_u32 {s(L_BY0M1)}, {s(S_T1)}, 1")
    E.i(f"s_sub_u32 {s(L_BX0M1)}, {s(S_T2)}, 1")
    for j in range(DMA_PER_WAVE):
        E.i(f"v_and_b32 {v(V_T0)}, 0xffff, {v(V_PYPX0 + j)}")
        E.i(f"v_lshrrev_b32 {v(V_T0 + 1)}, 16, {v(V_PYPX0 + j)}")
        E.i(f"v_add_u32 {v(V_T0)}, {s(L_BY0M1)}, {v(V_T0)}")
        E.i(f"v_add_u32 {v(V_T0 + 1)}, {s(L_BX0M1)}, {v(V_T0 + 1)}")
        E.i(f"v_cmp_gt_u32 vcc, {s(A_H)}, {v(V_T0)}")
        E.i(f"v_cmp_gt_u32 {s(S_T3, 2)}, {s(A_W)}, {v(V_T0 + 1)}")
        E.i(f"s_and_b64 vcc, vcc, {s(S_T3, 2)}")
        E.i(f"v_add_u32 {v(V_T0)}, {s(L_BASEPIX)}, {v(V_REL0 + j)}")
        E.i(f"v_cndmask_b32 {v(V_VOFF0 + j)}, -1, {v(V_T0)}, vcc")


def emit_dma_chunk(E, buf):
    for j in range(DMA_PER_WAVE):
        E.i(f"s_add_u32 m0, {s(L_M0BASE)}, {buf * RAWBUF_B + 4096 * j}")
        E.i("s_nop 0")
        E.i(f"buffer_load_dwordx4 {v(V_VOFF0 + j)}, {s(R_IN, 4)}, {s(L_DMAOFF)} offen lds")


def emit_u_ring_fill(E):
    for j in range(UD):
        E.i(f"buffer_load_dwordx4 {v(U0 + 4 * j, 4)}, {v(V_UVOFF)}, {s(R_U, 4)}, {s(L_UOFF)} offen")
        E.i(f"s_add_u32 {s(L_UOFF)}, {s(L_UOFF)}, {s(A_UPOS)}")


# --------------------------------------------------------------------------------------------------------------- epilogue
def emit_epilogue(E, pool):
    """Y = A^T M A in-lane; unit = (tile group, pair of tile columns); lane = (channel j16 of the wave's 16, tile row kq)"""
    E.c("epilogue" + (" + pooling" if pool else ""))
    E.i(f"v_mov_b32 {v(EP_VX)}, {v(V_OUTOFF)}")
    for x in range(1, 16):
        E.i(f"v_add_u32 {v(EP_VX + x)}, {s(A_PIXOUT)}, {v(EP_VX + x - 1)}")
    E.i(f"s_mul_i32 {s(S_T0)}, {s(A_W)}, {s(A_PIXOUT)}")
    E.i(f"s_mov_b32 {s(S_ROW)}, {s(C_TOUT)}")
    for i in range(1, 4):
        E.i(f"s_add_u32 {s(S_ROW + i)}, {s(S_ROW + i - 1)}, {s(S_T0)}")
    E.i(f"s_lshl_b32 {s(S_G1)}, {s(A_PIXOUT)}, 4")                           # 16 pixels to the right
    if pool:
        E.i(f"v_mov_b32 {v(EP_VXP)}, {v(V_POOLOFF)}")
        for x in range(1, 8):
            E.i(f"v_add_u32 {v(EP_VXP + x)}, {s(A_PIXPOOL)}, {v(EP_VXP + x - 1)}")
        E.i(f"s_lshr_b32 {s(S_T0)}, {s(A_W)}, 1")
        E.i(f"s_mul_i32 {s(S_T0)}, {s(S_T0)}, {s(A_PIXPOOL)}")
        E.i(f"s_mov_b32 {s(S_PROW)}, {s(C_TPOOL)}")
        E.i(f"s_add_u32 {s(S_PROW + 1)}, {s(S_PROW)}, {s(S_T0)}")
        E.i(f"s_lshl_b32 {s(S_PG1)}, {s(A_PIXPOOL)}, 3")                     # 8 pooled pixels to the right

    def pair(reg):
        return v(reg, 2)

    def first_pass(m, dst_t, stride):
        s12, d12, s34, d34 = EP_S, EP_S + 2, EP_S + 4, EP_S + 6
        E.i(f"v_pk_add_f32 {pair(s12)}, {pair(m[1])}, {pair(m[2])}")
        E.i(f"v_pk_add_f32 {pair(d12)}, {pair(m[1])}, {pair(m[2])} neg_lo:[0,1] neg_hi:[0,1]")
        E.i(f"v_pk_add_f32 {pair(s34)}, {pair(m[3])}, {pair(m[4])}")
        E.i(f"v_pk_add_f32 {pair(d34)}, {pair(m[3])}, {pair(m[4])} neg_lo:[0,1] neg_hi:[0,1]")
        t = [dst_t + stride * i for i in range(4)]
        E.i(f"v_pk_add_f32 {pair(t[0])}, {pair(m[0])}, {pair(s12)}")
        E.i(f"v_pk_add_f32 {pair(t[0])}, {pair(t[0])}, {pair(s34)}")
        E.i(f"v_pk_fma_f32 {pair(t[1])}, {s(K_2, 2)}, {pair(d34)}, {pair(d12)}")
        E.i(f"v_pk_fma_f32 {pair(t[2])}, {s(K_4, 2)}, {pair(s34)}, {pair(s12)}")
        E.i(f"v_pk_fma_f32 {pair(t[3])}, {s(K_8, 2)}, {pair(d34)}, {pair(d12)}")
        E.i(f"v_pk_add_f32 {pair(t[3])}, {pair(t[3])}, {pair(m[5])}")

    for g in range(2):
        for h2 in range(2):
            r0 = 2 * h2
            for nu in range(6):
                m = []
                for xi in range(6):
                    p = 6 * xi + nu
                    if p < 32:
                        dst = EP_M + 2 * xi
                        E.i(f"v_accvgpr_read_b32 {v(dst)}, {a(8 * p + 4 * g + r0)}")
                        E.i(f"v_accvgpr_read_b32 {v(dst + 1)}, {a(8 * p + 4 * g + r0 + 1)}")
                        m.append(dst)
                    else:
                        m.append(ACCV + 8 * (p - 32) + 4 * g + r0)
                first_pass(m, EP_T + 2 * nu, 12)
            for i in range(4):
                t = [EP_T + 2 * (6 * i + nu) for nu in range(6)]
                first_pass(t, EP_Y, 2)
                for k in range(4):
                    for rr in range(2):
                        E.i(f"v_max_f32 {v(EP_Y + 2 * k + rr)}, {s(A_RELU)}, {v(EP_Y + 2 * k + rr)}")
                row_s = S_ROW + i
                if g == 1:
                    E.i(f"s_add_u32 {s(S_T1)}, {s(S_ROW + i)}, {s(S_G1)}")
                    row_s = S_T1
                for k in range(4):
                    for rr in range(2):
                        x = 4 * (r0 + rr) + k
                        E.i(f"buffer_store_dword {v(EP_Y + 2 * k + rr)}, {v(EP_VX + x)}, {s(R_OUT, 4)}, {s(row_s)} offen" + (" " + ST_POLICY if ST_POLICY else ""))
                if pool:
                    if i % 2 == 0:
                        for rr in range(2):
                            E.i(f"v_max_f32 {v(EP_C + rr)}, {v(EP_Y + rr)}, {v(EP_Y + 2 + rr)}")
                            E.i(f"v_max_f32 {v(EP_C + 2 + rr)}, {v(EP_Y + 4 + rr)}, {v(EP_Y + 6 + rr)}")
                    else:
                        for rr in range(2):
                            E.i(f"v_max3_f32 {v(EP_C + rr)}, {v(EP_Y + rr)}, {v(EP_Y + 2 + rr)}, {v(EP_C + rr)}")
                            E.i(f"v_max3_f32 {v(EP_C + 2 + rr)}, {v(EP_Y + 4 + rr)}, {v(EP_Y + 6 + rr)}, {v(EP_C + 2 + rr)}")
                        prow_s = S_PROW + (i >> 1)
                        if g == 1:
                            E.i(f"s_add_u32 {s(S_T1)}, {s(S_PROW + (i >> 1))}, {s(S_PG1)}")
                            prow_s = S_T1
                        for rr in range(2):
                            for j in range(2):
                                xp = 2 * (r0 + rr) + j
                                E.i(f"buffer_store_dword {v(EP_C + 2 * j + rr)}, {v(EP_VXP + xp)}, {s(R_POOL, 4)}, {s(prow_s)} offen" + (" " + ST_POLICY if ST_POLICY else ""))


# ----------------------------------------------------------------------------------------------------------------- kernel
def emit_kernel(E, name):
    E.lines += [
        '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"',
        "\t.amdhsa_code_object_version 6",
        "\t.text",
        f"\t.protected\t{name}",
        f"\t.globl\t{name}",
        "\t.p2align\t8",
        f"\t.type\t{name},@function",
        f"{name}:",
    ]
    TID, LANE, J16, KQ, TT, TQ, TMPA, TMPB = V_T0, V_T0 + 1, V_T0 + 2, V_T0 + 3, V_T0 + 4, V_T0 + 5, V_T0 + 6, V_T0 + 7
    E.i(f"s_load_dwordx16 {s(8, 16)}, {s(S_KARG, 2)}, 0x0")
    E.i(f"s_load_dwordx16 {s(24, 16)}, {s(S_KARG, 2)}, 0x40")
    E.i(f"v_and_b32 {v(TID)}, 0x3ff, v0")
    E.i(f"v_lshrrev_b32 {v(TMPA)}, 6, {v(TID)}")
    E.i("s_nop 1")                                                            # VALU-written VGPR -> v_readfirstlane: one wait state (gfx940+)
    E.i(f"v_readfirstlane_b32 {s(L_WAVE)}, {v(TMPA)}")
    E.i("s_nop 1")
    E.i(f"v_and_b32 {v(LANE)}, 63, {v(TID)}")
    E.i(f"v_and_b32 {v(J16)}, 15, {v(LANE)}")
    E.i(f"v_lshrrev_b32 {v(KQ)}, 4, {v(LANE)}")
    E.i(f"v_lshrrev_b32 {v(TT)}, 2, {v(LANE)}")
    E.i(f"v_and_b32 {v(TQ)}, 3, {v(LANE)}")
    # ---- MFMA-role lane constants: V fragment of the group's tile j16, k-quad kq; U fragment of channel 16 wave + j16
    E.i(f"v_lshrrev_b32 {v(TMPA)}, 1, {v(J16)}")
    E.i(f"v_and_b32 {v(TMPA)}, 2, {v(TMPA)}")
    E.i(f"v_xor_b32 {v(TMPA)}, {v(TMPA)}, {v(KQ)}")
    E.i(f"v_lshlrev_b32 {v(TMPA)}, 4, {v(TMPA)}")
    E.i(f"v_lshl_or_b32 {v(V_VRD0)}, {v(J16)}, 6, {v(TMPA)}")
    E.i(f"v_add_u32 {v(V_VRD1)}, {32 * VPOS_B}, {v(V_VRD0)}")
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(L_WAVE)}, 10")                            # wave * 16 channels * 64 bytes
    E.i(f"v_lshlrev_b32 {v(TMPA)}, 4, {v(KQ)}")
    E.i(f"v_lshl_or_b32 {v(TMPA)}, {v(J16)}, 6, {v(TMPA)}")
    E.i(f"v_add_u32 {v(V_UVOFF)}, {s(S_T0)}, {v(TMPA)}")
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(L_WAVE)}, 6")                             # wave * 16 channels * 4 bytes
    E.i(f"v_lshlrev_b32 {v(TMPA)}, 2, {v(J16)}")
    E.i(f"v_add_u32 {v(V_CHAN4)}, {s(S_T0)}, {v(TMPA)}")
    # ---- transform lane constants.  Wave w: two-row piece (rows 1,2 for even w: alpha -4, beta 1; rows 3,4 for odd w: alpha -1, beta 2)
    # of tile group w >> 1, one-row piece (row 0 for even w, row 5 for odd w) of the OTHER group.  Patch rows: two-row pieces read
    # rows 1..4 (row0 = 1, step 1), row 0 reads rows 0,2,4 (row0 = 0, step 2), row 5 reads rows 1,3,5 (row0 = 1, step 2).
    E.i(f"v_lshrrev_b32 {v(TMPA)}, 2, {v(TT)}")                              # ty
    E.i(f"v_and_b32 {v(TMPB)}, 3, {v(TT)}")                                  # tx (inside the group)
    E.i(f"v_mul_u32_u24 {v(V_PRD)}, {4 * ROWP * 64}, {v(TMPA)}")
    E.i(f"v_lshl_add_u32 {v(V_PRD)}, {v(TMPB)}, 6, {v(V_PRD)}")
    E.i(f"v_lshl_add_u32 {v(V_PRD)}, {v(TQ)}, 4, {v(V_PRD)}")                # (4 ty rows, tx, quad) of the lane
    E.i(f"s_lshr_b32 {s(S_T1)}, {s(L_WAVE)}, 1")                             # g1 = w >> 1
    E.i(f"s_and_b32 {s(S_T2)}, {s(L_WAVE)}, 1")                              # odd
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(S_T1)}, 8")                               # g1 * 4 slots * 64 bytes
    E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {LDS_R0 + ROWP * 64}")            # + row 1
    E.i(f"s_xor_b32 {s(S_T3)}, {s(S_T1)}, 1")                                # g2 = 1 - g1
    E.i(f"s_lshl_b32 {s(S_T4)}, {s(S_T3)}, 8")
    E.i(f"s_mul_i32 {s(S_T3)}, {s(S_T2)}, {ROWP * 64}")                      # row0 of the one-row piece = odd
    E.i(f"s_add_u32 {s(S_T4)}, {s(S_T4)}, {s(S_T3)}")
    E.i(f"s_add_u32 {s(S_T4)}, {s(S_T4)}, {LDS_R0}")
    E.i(f"v_add_u32 {v(V_PRD2)}, {s(S_T4)}, {v(V_PRD)}")
    E.i(f"v_add_u32 {v(V_PRD)}, {s(S_T0)}, {v(V_PRD)}")
    # V write bases: position row xi_a of the piece, tile 16 g + t, quad ^ swz(t)
    E.i(f"v_and_b32 {v(TMPA)}, 1, {v(TMPA)}")
    E.i(f"v_lshlrev_b32 {v(TMPA)}, 1, {v(TMPA)}")                            # swz(tile)
    E.i(f"v_xor_b32 {v(TMPA)}, {v(TMPA)}, {v(TQ)}")
    E.i(f"v_lshlrev_b32 {v(TMPA)}, 4, {v(TMPA)}")
    E.i(f"v_lshl_or_b32 {v(TMPA)}, {v(TT)}, 6, {v(TMPA)}")
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(S_T2)}, 1")
    E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, 1")                                # xi_a = 1 (even) or 3 (odd)
    E.i(f"s_mul_i32 {s(S_T0)}, {s(S_T0)}, {6 * VPOS_B}")
    E.i(f"s_lshl_b32 {s(S_T3)}, {s(S_T1)}, 10")                              # g1 * 16 tiles * 64 bytes
    E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_T3)}")
    E.i(f"v_add_u32 {v(V_VWR)}, {s(S_T0)}, {v(TMPA)}")
    E.i(f"s_mul_i32 {s(S_T0)}, {s(S_T2)}, {5 * 6 * VPOS_B}")                 # xi = 0 (even) or 5 (odd)
    E.i(f"s_xor_b32 {s(S_T3)}, {s(S_T1)}, 1")
    E.i(f"s_lshl_b32 {s(S_T3)}, {s(S_T3)}, 10")
    E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_T3)}")
    E.i(f"v_add_u32 {v(V_VWR2)}, {s(S_T0)}, {v(TMPA)}")

    def kpair(reg, value):
        E.i(f"s_mov_b32 {s(reg)}, {value}")
        E.i(f"s_mov_b32 {s(reg + 1)}, {value}")
    kpair(K_4, "4.0"); kpair(K_M5, "0xc0a00000"); kpair(K_2, "2.0"); kpair(K_M2, "-2.0"); kpair(K_M4, "-4.0"); kpair(K_8, "0x41000000")
    E.i(f"s_bitcmp0_b32 {s(L_WAVE)}, 0")                                       # even wave: rows 1, 2
    for reg, v0_, v1_ in ((K_ALPHA, "-4.0", "-1.0"), (K_BETA, "1.0", "2.0"), (K_MBETA, "-1.0", "-2.0")):
        E.i(f"s_cselect_b32 {s(reg)}, {v0_}, {v1_}")
        E.i(f"s_mov_b32 {s(reg + 1)}, {s(reg)}")
    E.i("s_waitcnt lgkmcnt(0)")
    E.checkpoint(1, "arguments loaded, lane constants of the two roles computed")
    # ---- LDS-DMA lane constants: load L = wave + 4 j (j = 0..10), slot g = 16 L + lane / 4 -> patch pixel (py, px)
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(L_WAVE)}, 10")
    E.i(f"s_add_u32 {s(L_M0BASE)}, {s(S_T0)}, {LDS_R0}")
    E.i(f"s_mov_b32 {s(S_T2)}, 1821")                                         # g / 36 = (g * 1821) >> 16 for g < 704
    for j in range(DMA_PER_WAVE):
        g, py, sl, a9, px = TMPA, TMPB, V_PYPX0 + j, V_REL0 + j, V_VOFF0 + j
        E.i(f"s_lshl_b32 {s(S_T0)}, {s(L_WAVE)}, 4")
        E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {64 * j}")
        E.i(f"v_add_u32 {v(g)}, {s(S_T0)}, {v(TT)}")
        E.i(f"v_mul_lo_u32 {v(py)}, {v(g)}, {s(S_T2)}")
        E.i(f"v_lshrrev_b32 {v(py)}, 16, {v(py)}")                           # g / 36
        E.i(f"v_mul_u32_u24 {v(sl)}, 36, {v(py)}")
        E.i(f"v_sub_u32 {v(sl)}, {v(g)}, {v(sl)}")                           # g % 36
        E.i(f"v_mul_u32_u24 {v(a9)}, 57, {v(sl)}")
        E.i(f"v_lshrrev_b32 {v(a9)}, 9, {v(a9)}")                            # c = sl / 9 (57 / 512)
        E.i(f"v_mul_u32_u24 {v(px)}, 9, {v(a9)}")
        E.i(f"v_sub_u32 {v(px)}, {v(sl)}, {v(px)}")                          # a = sl % 9
        E.i(f"v_lshl_add_u32 {v(px)}, {v(px)}, 2, {v(a9)}")                  # pixel column 4 a + c
        E.i(f"v_cmp_gt_u32 vcc, {RAW_SLOTS}, {v(g)}")
        E.i(f"v_cmp_gt_u32 {s(S_T3, 2)}, 34, {v(px)}")
        E.i(f"s_and_b64 vcc, vcc, {s(S_T3, 2)}")
        E.i(f"v_add_u32 {v(g)}, -1, {v(py)}")
        E.i(f"v_mul_lo_u32 {v(g)}, {v(g)}, {s(A_W)}")
        E.i(f"v_add_u32 {v(g)}, {v(g)}, {v(px)}")
        E.i(f"v_add_u32 {v(g)}, -1, {v(g)}")
        E.i(f"v_mul_lo_u32 {v(g)}, {v(g)}, {s(A_PIXIN)}")
        E.i(f"v_lshl_add_u32 {v(V_REL0 + j)}, {v(TQ)}, 4, {v(g)}")
        E.i(f"v_lshl_or_b32 {v(py)}, {v(px)}, 16, {v(py)}")
        E.i(f"v_mov_b32 {v(g)}, 0x7fff")
        E.i(f"v_cndmask_b32 {v(V_PYPX0 + j)}, {v(g)}, {v(py)}, vcc")
    # ---- output lane constants
    E.i(f"s_lshl_b32 {s(S_T0)}, {s(A_W)}, 2")
    E.i(f"s_mul_i32 {s(S_T0)}, {s(S_T0)}, {s(A_PIXOUT)}")
    E.i(f"v_mul_lo_u32 {v(TMPA)}, {v(KQ)}, {s(S_T0)}")
    E.i(f"v_add_u32 {v(TMPA)}, {v(TMPA)}, {v(V_CHAN4)}")
    E.i(f"v_add_u32 {v(V_OUTOFF)}, {s(A_COOFF)}, {v(TMPA)}")
    E.i(f"s_mul_i32 {s(S_T0)}, {s(A_W)}, {s(A_PIXPOOL)}")
    E.i(f"v_mul_lo_u32 {v(TMPA)}, {v(KQ)}, {s(S_T0)}")
    E.i(f"v_add_u32 {v(V_POOLOFF)}, {v(TMPA)}, {v(V_CHAN4)}")
    E.i(f"s_mov_b32 {s(R_U)}, {s(A_U)}")
    E.i(f"s_and_b32 {s(R_U + 1)}, {s(A_U + 1)}, 0xffff")
    E.i(f"s_mov_b32 {s(R_U + 2)}, {s(A_UBYTES)}")
    E.i(f"s_mov_b32 {s(R_U + 3)}, 0x00020000")
    E.i(f"s_mov_b32 {s(R_IN + 2)}, {s(A_IMGIN)}")
    E.i(f"s_mov_b32 {s(R_IN + 3)}, 0x00020000")
    E.i(f"s_mov_b32 {s(R_OUT + 2)}, {s(A_IMGOUT)}")
    E.i(f"s_mov_b32 {s(R_OUT + 3)}, 0x00020000")
    E.i(f"s_mov_b32 {s(R_POOL + 2)}, {s(A_IMGPOOL)}")
    E.i(f"s_mov_b32 {s(R_POOL + 3)}, 0x00020000")
    # ---- this workgroup's share of its XCD's logical block range
    E.i(f"s_and_b32 {s(S_T0)}, {s(S_WG)}, 7")
    E.i(f"s_lshr_b32 {s(L_TT)}, {s(S_WG)}, 3")
    E.i(f"s_lshr_b32 {s(L_SLOTS)}, {s(A_GRID)}, 3")
    E.i(f"s_and_b32 {s(S_T1)}, {s(A_GRID)}, 7")
    E.i(f"s_cmp_lt_u32 {s(S_T0)}, {s(S_T1)}")
    E.i(f"s_addc_u32 {s(L_SLOTS)}, {s(L_SLOTS)}, 0")
    E.i(f"s_lshr_b32 {s(S_T1)}, {s(A_NWG)}, 3")
    E.i(f"s_and_b32 {s(S_T2)}, {s(A_NWG)}, 7")
    E.i(f"s_min_u32 {s(S_T3)}, {s(S_T0)}, {s(S_T2)}")
    E.i(f"s_mul_i32 {s(L_TSTART)}, {s(S_T0)}, {s(S_T1)}")
    E.i(f"s_add_u32 {s(L_TSTART)}, {s(L_TSTART)}, {s(S_T3)}")
    E.i(f"s_cmp_lt_u32 {s(S_T0)}, {s(S_T2)}")
    E.i(f"s_addc_u32 {s(L_TCOUNT)}, {s(S_T1)}, 0")
    E.i(f"s_cmp_ge_u32 {s(L_TT)}, {s(L_TCOUNT)}")
    E.i("s_cbranch_scc1 .Lend_program")
    E.i(f"s_add_u32 {s(S_T0)}, {s(L_TSTART)}, {s(L_TT)}")
    emit_setup_tile(E)
    E.checkpoint(3, "first block decoded, its bias load issued")
    E.i(f"s_mov_b32 {s(N_HAS)}, 1")
    E.i(f"s_mov_b32 {s(L_UOFF)}, {s(N_UBASE)}")
    E.i(f"s_mov_b32 {s(L_DMAOFF)}, 0")
    emit_dma_chunk(E, 0)
    E.i(f"s_mov_b32 {s(L_DMAOFF)}, 64")
    emit_dma_chunk(E, 1)
    E.i(f"s_mov_b32 {s(L_DMAOFF)}, 128")
    emit_u_ring_fill(E)
    E.checkpoint(5, "first two raw chunks and the U ring requested")
    E.i("s_waitcnt vmcnt(0)")
    for R in ("A",):                                                          # one instruction stream for the four waves
        E.i(f"s_branch .Ljoin_{R}")
        # ================================================================================================ tile loop
        E.label(f".Ltile_{R}")
        emit_body(E, 0, "first")
        E.checkpoint(8, "first chunk done")
        E.label(f".Lloop_{R}")
        emit_body(E, 1, "mid")
        E.i(f"s_cmp_eq_u32 {s(L_PAIRS)}, 1")
        E.i(f"s_cbranch_scc0 .Lnosetup_{R}")
        # ---- the next block: from here on the LDS-DMA loads fetch ITS first two chunks
        E.i(f"s_add_u32 {s(S_T0)}, {s(L_TT)}, {s(L_SLOTS)}")
        E.i(f"s_mov_b32 {s(L_DMAOFF)}, 0")
        E.i(f"s_mov_b32 {s(N_HAS)}, 0")
        E.i(f"s_cmp_lt_u32 {s(S_T0)}, {s(L_TCOUNT)}")
        E.i(f"s_cbranch_scc0 .Lnosetup_{R}")
        E.i(f"s_mov_b32 {s(N_HAS)}, 1")
        E.i(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(L_TSTART)}")
        emit_setup_tile(E)
        E.label(f".Lnosetup_{R}")
        emit_body(E, 0, "mid")
        E.i(f"s_sub_u32 {s(L_PAIRS)}, {s(L_PAIRS)}, 1")
        E.i(f"s_cmp_lg_u32 {s(L_PAIRS)}, 0")
        E.i(f"s_cbranch_scc1 .Lloop_{R}")
        emit_body(E, 1, "last")
        E.checkpoint(9, "every chunk of the block done; epilogue next")
        # every load that opens the next block (its U ring, raw chunks 0 and 1, bias) is home BEFORE the epilogue's stores join the queue
        E.i("s_waitcnt vmcnt(0)")
        E.i("s_nop 7")
        E.i("s_nop 7")
        E.i("s_branch .Lepilogue")
        E.label(f".Lepi_done_{R}")
        E.i(f"s_cmp_eq_u32 {s(N_HAS)}, 0")
        E.i("s_cbranch_scc1 .Lend_program")
        E.i(f"s_add_u32 {s(L_TT)}, {s(L_TT)}, {s(L_SLOTS)}")
        E.label(f".Ljoin_{R}")
        E.i(f"s_mov_b32 {s(R_OUT)}, {s(N_OUTB)}")
        E.i(f"s_mov_b32 {s(R_OUT + 1)}, {s(N_OUTB + 1)}")
        E.i(f"s_mov_b32 {s(R_POOL)}, {s(N_POOLB)}")
        E.i(f"s_mov_b32 {s(R_POOL + 1)}, {s(N_POOLB + 1)}")
        E.i(f"s_mov_b32 {s(C_TOUT)}, {s(N_TOUT)}")
        E.i(f"s_mov_b32 {s(C_TPOOL)}, {s(N_TPOOL)}")
        E.i(f"s_lshr_b32 {s(L_PAIRS)}, {s(A_NCH)}, 1")
        E.i(f"s_sub_u32 {s(L_PAIRS)}, {s(L_PAIRS)}, 1")
        for g in range(2):                                                   # bias = the initial value of position (1, 1)
            for r in range(4):
                E.i(f"v_accvgpr_write_b32 {a(8 * 7 + 4 * g + r)}, {v(V_BIAS0)}")
        E.checkpoint(7, "joined: bias in the accumulators")
        E.i(f"s_branch .Ltile_{R}")
    E.label(".Lepilogue")
    E.i(f"s_bitcmp1_b32 {s(A_FLAGS)}, 0")
    E.i("s_cbranch_scc0 .Lepi_nopool")
    emit_epilogue(E, True)
    E.i("s_branch .Lepi_return")
    E.label(".Lepi_nopool")
    emit_epilogue(E, False)
    E.label(".Lepi_return")
    E.i("s_branch .Lepi_done_A")
    E.label(".Lend_program")
    E.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E.i("s_endpgm")
    E.lines += ["\t.p2alignl 6, 3212836864", "\t.fill 256, 4, 3212836864"]
    E.lines += [
        "\t.section\t.rodata,\"a\",@progbits",
        "\t.p2align\t6, 0x0",
        f"\t.amdhsa_kernel {name}",
        f"\t\t.amdhsa_group_segment_fixed_size {LDS_BYTES}",
        "\t\t.amdhsa_private_segment_fixed_size 0",
        "\t\t.amdhsa_kernarg_size 128",
        "\t\t.amdhsa_user_sgpr_count 2",
        "\t\t.amdhsa_user_sgpr_dispatch_ptr 0",
        "\t\t.amdhsa_user_sgpr_queue_ptr 0",
        "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1",
        "\t\t.amdhsa_user_sgpr_dispatch_id 0",
        "\t\t.amdhsa_user_sgpr_kernarg_preload_length 0",
        "\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0",
        "\t\t.amdhsa_user_sgpr_private_segment_size 0",
        "\t\t.amdhsa_uses_dynamic_stack 0",
        "\t\t.amdhsa_enable_private_segment 0",
        "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
        "\t\t.amdhsa_system_sgpr_workgroup_id_y 0",
        "\t\t.amdhsa_system_sgpr_workgroup_id_z 0",
        "\t\t.amdhsa_system_sgpr_workgroup_info 0",
        "\t\t.amdhsa_system_vgpr_workitem_id 0",
        "\t\t.amdhsa_next_free_vgpr 512",
        "\t\t.amdhsa_next_free_sgpr 102",
        "\t\t.amdhsa_accum_offset 256",
        "\t\t.amdhsa_reserve_vcc 1",
        "\t\t.amdhsa_float_round_mode_32 0",
        "\t\t.amdhsa_float_round_mode_16_64 0",
        "\t\t.amdhsa_float_denorm_mode_32 3",
        "\t\t.amdhsa_float_denorm_mode_16_64 3",
        "\t\t.amdhsa_dx10_clamp 1",
        "\t\t.amdhsa_ieee_mode 1",
        "\t\t.amdhsa_fp16_overflow 0",
        "\t\t.amdhsa_tg_split 0",
        "\t\t.amdhsa_exception_fp_ieee_invalid_op 0",
        "\t\t.amdhsa_exception_fp_denorm_src 0",
        "\t\t.amdhsa_exception_fp_ieee_div_zero 0",
        "\t\t.amdhsa_exception_fp_ieee_overflow 0",
        "\t\t.amdhsa_exception_fp_ieee_underflow 0",
        "\t\t.amdhsa_exception_fp_ieee_inexact 0",
        "\t\t.amdhsa_exception_int_div_zero 0",
        "\t.end_amdhsa_kernel",
        "\t.text",
        "\t.amdgpu_metadata",
        "---",
        "amdhsa.kernels:",
        "  - .agpr_count:     256",
        "    .args:",
        "      - .offset:         0",
        "        .size:           128",
        "        .value_kind:     by_value",
        f"    .group_segment_fixed_size: {LDS_BYTES}",
        "    .kernarg_segment_align: 8",
        "    .kernarg_segment_size: 128",
        "    .max_flat_workgroup_size: 256",
        f"    .name:           {name}",
        "    .private_segment_fixed_size: 0",
        "    .sgpr_count:     108",
        "    .sgpr_spill_count: 0",
        f"    .symbol:         {name}.kd",
        "    .uniform_work_group_size: 1",
        "    .uses_dynamic_stack: false",
        "    .vgpr_count:     512",
        "    .vgpr_spill_count: 0",
        "    .wavefront_size: 64",
        "amdhsa.target:   amdgcn-amd-amdhsa--gfx950",
        "amdhsa.version:",
        "  - 1",
        "  - 2",
        "...",
        "",
        "\t.end_amdgpu_metadata",
    ]


def main():
    global STOP_AT, DUMP, DMA_POS, ST_POLICY
    out = sys.argv[1] if len(sys.argv) > 1 else "wino4b_gfx950.s"
    if "--ud" in sys.argv:
        layout(int(sys.argv[sys.argv.index("--ud") + 1]))
    if "--dma-pos" in sys.argv:
        DMA_POS = tuple(int(x) for x in sys.argv[sys.argv.index("--dma-pos") + 1].split(","))
        assert len(DMA_POS) == DMA_PER_WAVE and len(set(DMA_POS)) == DMA_PER_WAVE and max(DMA_POS) < 36
    if "--store-policy" in sys.argv:
        ST_POLICY = sys.argv[sys.argv.index("--store-policy") + 1]
    if "--dump" in sys.argv:
        DUMP = sys.argv[sys.argv.index("--dump") + 1]
    if "--stop" in sys.argv:
        STOP_AT = int(sys.argv[sys.argv.index("--stop") + 1])
    for g in range(704):
        assert (g * 1821) >> 16 == g // 36, g
    for sl in range(36):
        assert (sl * 57) >> 9 == sl // 9, sl
    assert 36 % UD == 0 and LDS_BYTES <= 163840
    E = E2()
    emit_kernel(E, "conv3x3_wino4b_f32")
    open(out, "w").write("\n".join(E.lines) + "\n")
    top = sorted(E.stats.items(), key=lambda kv: -kv[1])[:10]
    print(f"{out}: {sum(E.stats.values())} instructions; " + ", ".join(f"{k} {n}" for k, n in top), file=sys.stderr)


if __name__ == "__main__":
    main()
