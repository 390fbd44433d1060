#!/bin/bash
# tools/var_py.sh <lib> <python script and arguments>: run a script on another build of the library (GPU box scratch copy)
lib=$1; shift
cp figaroh_plus_amd/libfigh.so /tmp/libfigh_shipped.so; cp $lib figaroh_plus_amd/libfigh.so
python "$@"; cp /tmp/libfigh_shipped.so figaroh_plus_amd/libfigh.so
