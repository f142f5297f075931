#!/bin/bash
# usage (GPU box): tools/pre2_ab.sh  -- source pre-pass one block per wavefront (2) against four (1): headline
cd "$GRAFT_REPO_ROOT"
tools/ab_env.sh DSV2_HME_PRESTATS=2 DSV2_HME_PRESTATS=1 DSV2_HME_PRESTATS=2 DSV2_HME_PRESTATS=1 2>&1 | cut -c1-330
