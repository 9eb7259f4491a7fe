#!/bin/bash
# Junction study (CPU oracle, STUDY BUILD, build container only): what two junction rules of SUMO that this model does not have would
# change in the reference-held delay cells.  -DRM_STUDY_JUNCTION bits: 1 link leaders (no entry while a MOVING vehicle of a conflicting
# movement -- <request foes>, crossing and merging, whatever the priority -- is on its junction lanes), 2 standing vehicles count too,
# 4 keepClear (no entry unless the vehicle fits behind the last standing vehicle of its destination lane).
#   python oracle/study/conflicts.py && bash oracle/study/junction_study.sh > profiles/r04_junction_study.txt
cd "$(dirname "$0")/../.." || exit 1
for v in 0 1 3 4 5 7; do
  if [ $v = 0 ]; then make -C oracle -B -s; echo "== shipped model"; python oracle/fidelity_eval.py --compact | tail -7
  else make -C oracle -B -s ORC_DEFS="-DRM_STUDY_JUNCTION=$v"; echo "== RM_STUDY_JUNCTION=$v"; ORC_STUDY_CONFLICTS=1 python oracle/fidelity_eval.py --compact | tail -7; fi
done
make -C oracle -B -s
