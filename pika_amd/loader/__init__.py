"""On-the-fly loader: .mrk/.seq audio + label archives on the host,
speed/volume perturbation, Kaldi-compatible fbank, splice and batching on the GPU."""
