# source me: run_bounded SECONDS cmd... -- runs cmd in its own process group and kills the whole group at the deadline
# (the watchdog's stdio is detached: an inherited pipe would keep `| tail` waiting until the deadline)
run_bounded() {
  local secs=$1; shift
  setsid "$@" &
  local p=$!
  ( sleep "$secs"; kill -KILL -- -"$p" 2>/dev/null ) </dev/null >/dev/null 2>&1 &
  local w=$!
  wait "$p"; local rc=$?
  pkill -P "$w" sleep 2>/dev/null; kill "$w" 2>/dev/null
  return $rc
}
