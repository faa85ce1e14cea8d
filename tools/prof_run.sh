#!/bin/bash
# rocprofv3 passes of bench.py for profiles/: kernel trace (+stats) and three PMC passes (separate, as MI355X_MICROARCH.md asks).
# usage: bash tools/prof_run.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}; summaries into profiles/
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --overlap-streams 0 --no-extras $@"
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --e2e-files 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_trace.json 2> $out/trace.err || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $P > $out/bench_fetch.json 2> $out/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $P > $out/bench_write.json 2> $out/write.err || echo "write failed"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -- $P > $out/bench_sq.json 2> $out/sq.err || echo "sq failed"
python3 profiles/summarize.py trace ${tag} $out/trace 2 10 > $out/summ_trace.log 2>&1
python3 profiles/summarize.py ${tag} $out/trace $out/fetch $out/write $out/sq > $out/summ_pmc.log 2>&1
python3 profiles/summarize.py trace ${tag} $out/trace 2 10 > $out/summ_trace.log 2>&1
# the reference's other two modes (p_ref_inp = None, i_reinterp = 1): k_ps_loop_multi<..., LOCAL> and k_reinterp_pair
M="python3 tools/modes_run.py $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mtrace -- $M > $out/modes_trace.json 2> $out/mtrace.err || echo "modes trace failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/mfetch -- $M > $out/modes_fetch.json 2> $out/mfetch.err || echo "modes fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/mwrite -- $M > $out/modes_write.json 2> $out/mwrite.err || echo "modes write failed"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/msq -- $M > $out/modes_sq.json 2> $out/msq.err || echo "modes sq failed"
python3 profiles/summarize.py ${tag}_modes $out/mtrace $out/mfetch $out/mwrite $out/msq > $out/summ_modes_pmc.log 2>&1
python3 profiles/summarize.py trace ${tag}_modes $out/mtrace 2 4 > $out/summ_modes_trace.log 2>&1
cp profiles/kernel_stats_${tag}.csv profiles/pmc_summary_${tag}.json profiles/kernel_stats_${tag}_modes.csv profiles/pmc_summary_${tag}_modes.json $out/ 2>/dev/null
# keep the merged-back directory small: per-dispatch counter tables are large
find $out -name "*counter_collection.csv" -size +3M -delete
find $out -name "*kernel_trace.csv" -size +3M -delete
