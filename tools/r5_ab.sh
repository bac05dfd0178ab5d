#!/bin/bash
# same-box A/B of library variants on opbench shapes: tools/r5_ab.sh "<variant names...>" -- <opbench args> [-- <opbench args> ...]
# variant "tree" = aicity_action_amd/lib/libmvit_hip.so, others = aicity_action_amd/lib/variants/libmvit_hip_<name>.so
root=${GRAFT_REPO_ROOT:-$(pwd)}
vars=$1; shift
cmds=(); cur=""
for a in "$@"; do if [ "$a" == "--" ]; then [ -n "$cur" ] && cmds+=("$cur"); cur=""; else cur="$cur $a"; fi; done
[ -n "$cur" ] && cmds+=("$cur")
for rep in 1 2; do
 for c in "${cmds[@]}"; do
  for v in $vars; do
    lib=$root/aicity_action_amd/lib/libmvit_hip.so; [ "$v" != "tree" ] && lib=$root/aicity_action_amd/lib/variants/libmvit_hip_$v.so
    printf "%-8s %s\n" "$v" "$(MVIT_HIP_LIB=$lib python3 $root/tools/opbench.py $c 2>/dev/null | tail -1)"
  done
 done
done
