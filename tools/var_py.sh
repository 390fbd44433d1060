#!/bin/bash
# tools/var_py.sh <lib> <python script and arguments>: run a script on another build of the library (GPU box scratch copy)
lib=$1; shift
keep=$(mktemp /tmp/libfigh_shipped.XXXXXX.so); cp figaroh_plus_amd/libfigh.so $keep
trap 'cp $keep figaroh_plus_amd/libfigh.so; rm -f $keep' EXIT  # (also on a failing or interrupted run)
cp $lib figaroh_plus_amd/libfigh.so
python "$@"
