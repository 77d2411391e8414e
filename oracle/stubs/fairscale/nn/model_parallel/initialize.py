def get_model_parallel_world_size(): return 1
def get_model_parallel_rank(): return 0
def get_model_parallel_src_rank(): return 0
def get_model_parallel_group(): return None
def get_data_parallel_world_size(): return 1
def get_data_parallel_rank(): return 0
def initialize_model_parallel(n=1, *a, **k): return None
def model_parallel_is_initialized(): return True
