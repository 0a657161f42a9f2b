/* Embeds the learned parameter blobs (params/*.bin, see params/NOTICE) into libefx_hip.so. */
    .section .rodata
    .balign 16
    .global efx_blob_bad256
efx_blob_bad256:
    .incbin "../params/bad256.bin"
    .balign 16
    .global efx_blob_bad512
efx_blob_bad512:
    .incbin "../params/bad512.bin"
    .balign 16
    .global efx_blob_hashsift256
efx_blob_hashsift256:
    .incbin "../params/hashsift256.bin"
    .balign 16
    .global efx_blob_hashsift512
efx_blob_hashsift512:
    .incbin "../params/hashsift512.bin"
    .section .note.GNU-stack,"",@progbits
