#!/bin/bash
O=$1
C3OPTS="${C3OPTS:-none}" STEPS=20 bash tools/r5_steps/c3ab.sh $O
