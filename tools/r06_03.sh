# round 6: (a) same-box A/B of the trunk's neighbour-poll forms in the chunk-3 step (in-tree: poll only where a check follows; uncond: HEAD's
# unconditional poll; always: poll + check in both steps; rolled: HEAD's codegen), (b) mixed-vs-exact gradient parity of the full training step,
# (c) the MFMA ceiling of this package re-measured at HEAD with an SMI power / clock trace, (d) power / clock trace under the trunk itself
export TMPDIR=/tmp O=gpurun_out; mkdir -p $O
bash tools/ab_variants.sh uncond always rolled 2>&1 | tee $O/r06c_ab_poll_forms.txt
timeout 600 python tools/grad_parity.py 64 23 2>&1 | grep "^{" | tee $O/r06c_grad_parity_b64.txt
timeout 300 python tools/grad_parity.py 8 2 2>&1 | grep "^{" | tee $O/r06c_grad_parity_b8.txt
python tools/dump_trunk_operands.py $O/trunk_acts.bin $O/trunk_weights.bin
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'; sleep 0.5; done ) > $O/r06c_mfma_ceiling_smi.log 2>&1 &
SMI=$!
CEIL_SECONDS=4 CEIL_SKIP_ORDER=1 timeout 600 tools/mfma_ceiling $O/trunk_acts.bin $O/trunk_weights.bin 2>&1 | while IFS= read -r line; do echo "t=$(date +%s.%N) $line"; done | tee $O/r06c_mfma_ceiling.txt
kill $SMI
rm -f $O/trunk_acts.bin $O/trunk_weights.bin
timeout 300 python tools/power_probe.py 2>&1 | tee $O/r06c_trunk_power_probe.txt
