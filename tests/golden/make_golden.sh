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
