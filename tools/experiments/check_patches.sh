#!/bin/bash
# Verify that every experiment patch applies to the tree of the commit tools/experiments/README.md names for it.
# Needs the git history (a build-container check; the GPU box gets a snapshot without .git).
set -u
cd "$(git rev-parse --show-toplevel)"
rc=0
check() {   # commit, patch, mode
  local wt; wt=$(mktemp -d)
  git worktree add -q "$wt" "$1" || { echo "cannot check out $1"; rc=1; return; }
  if [ "$3" = csrc ]; then
    (cd "$wt/sound_event_detection_dcase2017_task4_amd/csrc" && patch -p1 --dry-run -s < "$OLDPWD/tools/experiments/$2" > /dev/null) && echo "ok   $2 @ $1" || { echo "FAIL $2 @ $1"; rc=1; }
  else
    (cd "$wt" && git apply --check "$OLDPWD/tools/experiments/$2") && echo "ok   $2 @ $1" || { echo "FAIL $2 @ $1"; rc=1; }
  fi
  git worktree remove --force "$wt"
}
check 12eacbc experiment_kernel_ablations.patch csrc
check 05b97d4 experiment_conv_tile_walk_prefetch.patch apply
check 0616006 experiment_pool_sums_in_dgrad_epilogue.patch apply
exit $rc
