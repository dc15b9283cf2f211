#!/bin/bash
# usage: isa_hist.sh file.s kernel_mangled_prefix  -> instruction histogram of one kernel
S=$1; K=$2
awk -v k="$K" 'index($0, k) == 1 && /:/ {p=1; next} p && /s_endpgm/ {p=0} p' "$S" | grep -E "^\s+[a-z]" | grep -vE "^\s+\." | awk '{print $1}' | sort | uniq -c | sort -rn | awk '{t+=$1; l=l" "$2":"$1} END{print "total", t; print l}'
