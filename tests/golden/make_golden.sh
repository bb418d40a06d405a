#!/bin/sh
# Regenerates tests/golden/*.vec from the UNMODIFIED reference compiled by oracle/Makefile
# (oracle/_ref/ref_tool; needs /root/reference + GMP).  Run from the repo root:
#     make -C oracle ref && sh tests/golden/make_golden.sh
set -e
T=oracle/_ref/ref_tool
A=pbc_amd/param/a.param
G=tests/golden
$T kat $A $G/a_kat.vec                        # pbc/pairing_test.pbc:3-10 known answer
$T gen $A chain  1024 1  1 $G/a_chain1024.vec # P_i=(i+1)P0, Q_i=(i+1)Q0 (SURVEY 8d inputs)
$T gen $A random 32   1 42 $G/a_rand32.vec    # element_random, seed 42
$T gen $A edge   20   1  7 $G/a_edge20.vec    # random + off-curve (-> O) inputs
$T gen $A random 4   16  5 $G/a_prod16x4.vec  # element_prod_pairing, 16 terms
$T gen $A chain  8    2  1 $G/a_prod2x8.vec
$T gen $A edge   10   3  9 $G/a_prod3x10_edge.vec
D=pbc_amd/param/d159.param
F=pbc_amd/param/f.param
$T gen $D chain  256 1  1 $G/d_chain256.vec
$T gen $D random 32  1 42 $G/d_rand32.vec
$T gen $D edge   20  1  7 $G/d_edge20.vec
$T gen $D random 4  16  5 $G/d_prod16x4.vec
$T gen $D edge   10  3  9 $G/d_prod3x10_edge.vec
$T gen $F chain  128 1  1 $G/f_chain128.vec
$T gen $F random 16  1 42 $G/f_rand16.vec
$T gen $F edge   10  1  7 $G/f_edge10.vec
$T gen $F random 3   4  5 $G/f_prod4x3.vec
$T gen $F edge   5   3  9 $G/f_prod3x5_edge.vec
$T hash $A 24 32 3 $G/a_hash32.vec              # element_from_hash(G1) on 32-byte digests
$T hash $A 8 13 4 $G/a_hash13.vec               # short input: H || 0 || H || 1 ... expansion (field.c:640-668)
$T hash $A 6 100 5 $G/a_hash100.vec             # long input: truncated to 64 bytes
# the other shipped type d parameter files: 175..224-bit q (6 / 7 words, 22..28-byte coordinates)
for d in d277699-175-167 d278027-190-181 d105171-196-185 d201 d224; do
  $T gen pbc_amd/param/$d.param random 12 1 42 $G/${d}_rand12.vec
  $T gen pbc_amd/param/$d.param edge   8  1  7 $G/${d}_edge8.vec
  $T gen pbc_amd/param/$d.param edge   4  3  9 $G/${d}_prod3x4_edge.vec
done
# type g (Freeman, k = 10), type a1 (1033-bit, composite order), type e (k = 1, 1020-bit)
$T gen pbc_amd/param/g149.param chain 64 1 1 $G/g149_chain64.vec
$T gen pbc_amd/param/g149.param random 16 1 42 $G/g149_rand16.vec
$T gen pbc_amd/param/g149.param edge 10 1 7 $G/g149_edge10.vec
$T gen pbc_amd/param/g149.param edge 4 3 9 $G/g149_prod3x4_edge.vec
$T gen pbc_amd/param/g149.param random 3 4 5 $G/g149_prod4x3.vec
for t in a1 e; do
  $T gen pbc_amd/param/$t.param random 6 1 42 $G/${t}_rand6.vec
  $T gen pbc_amd/param/$t.param edge 6 1 7 $G/${t}_edge6.vec
  $T gen pbc_amd/param/$t.param edge 3 3 9 $G/${t}_prod3x3_edge.vec
  $T gen pbc_amd/param/$t.param chain 8 1 1 $G/${t}_chain8.vec
done
# element_from_hash(G1) for the other families (Tonelli-Shanks fields, cofactors, no cofactor for f)
for pf in d159 d201 d278027-190-181 f g149; do $T hash pbc_amd/param/$pf.param 8 32 3 $G/${pf}_hash32.vec; done
$T hash pbc_amd/param/e.param 3 20 3 $G/e_hash20.vec
$T hash pbc_amd/param/a1.param 3 20 3 $G/a1_hash20.vec
# element_mul_zn on G2 (twists over F_q^d / F_q^2) and on G1 of wide fields
for pf in d159 d201 g149 f; do $T gmul pbc_amd/param/$pf.param 2 6 11 $G/${pf}_g2mul6.vec; done
$T gmul pbc_amd/param/e.param 1 3 11 $G/e_g1mul3.vec
$T gmul pbc_amd/param/d224.param 1 6 11 $G/d224_g1mul6.vec
# element_to_bytes_compressed / element_from_bytes_compressed on G1
for pf in a d159 d278027-190-181 f g149; do $T compress pbc_amd/param/$pf.param 12 21 $G/${pf}_compress12.vec; done
$T compress pbc_amd/param/e.param 4 21 $G/e_compress4.vec
