# A/B of two builds of the library in ONE session (clocks drift between sessions): stardis_amd/lib_old/ against stardis_amd/lib/,
# alternating.  The other build goes to stardis_amd/lib_old/libstardis_hip.so first (e.g. a copy of csrc/ with the file under test taken
# from `git show REV:path`, `make -C` there, cp of its ../lib/libstardis_hip.so); *.so files are not tracked but travel to the GPU box.
# usage: bash scripts/ab_libs.sh "<command>"
cp stardis_amd/lib/libstardis_hip.so /tmp/new.so
for round in 1 2; do
  cp stardis_amd/lib_old/libstardis_hip.so stardis_amd/lib/libstardis_hip.so; echo "== old ($round)"; eval "$1"
  cp /tmp/new.so stardis_amd/lib/libstardis_hip.so; echo "== new ($round)"; eval "$1"
done
