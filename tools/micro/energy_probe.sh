# usage (GPU box): bash tools/micro/energy_probe.sh   -> one JSON line per mode with the sustained rate, clock and socket power
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 energy_probe.hip -o /tmp/energy_probe 2>/dev/null || exit 1
for mode in ${MODES:-0 1 4 2 3}; do
  (timeout 40 /tmp/energy_probe $mode > /tmp/ep_$mode.txt 2>&1 &)
  sleep ${WARM:-3.5}
  smi=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' ')
  sleep ${TAIL:-2.5}
  echo "mode $mode | $(tail -1 /tmp/ep_$mode.txt) | $smi"
done
