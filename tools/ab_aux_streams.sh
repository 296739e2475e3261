python tools/ab_step.py dreg_exec_set_aux_streams 1 2 3 --rounds 3 --steps 12
