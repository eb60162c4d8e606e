from .pdb_writer import TrajectoryWriter, format_pdb, index_to_str_seq, postprocess_trajectory, save_pdb  # noqa: F401
