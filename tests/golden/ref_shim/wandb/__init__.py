# empty stand-in: util/misc.py imports wandb at module scope
