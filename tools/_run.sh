bash tools/final_measure.sh r3final
