#!/bin/bash
# tile form (MX_EQ_SPEC_SB) x submission length at 1024 strips, and the rank shapes: which form where?
cd $GRAFT_REPO_ROOT
for T in 64 128 256 512 1024; do
  for sb in 32 321; do MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --ticks $T 2>&1 | grep strips | sed 's/overlap=False //; s/fast=False //; s/| spec.*=>/=>/'; done
  for sb in 16 321; do MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --fp-contract --ticks $T 2>&1 | grep strips | sed 's/overlap=False //; s/fast=False //; s/| spec.*=>/=>/'; done
done
for S in 128 256 512; do
  for sb in 32 321; do MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --strips $S 2>&1 | grep strips | sed 's/overlap=False //; s/fast=False //; s/| spec.*=>/=>/'; done
  for sb in 16 321; do MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --fp-contract --strips $S 2>&1 | grep strips | sed 's/overlap=False //; s/fast=False //; s/| spec.*=>/=>/'; done
done
