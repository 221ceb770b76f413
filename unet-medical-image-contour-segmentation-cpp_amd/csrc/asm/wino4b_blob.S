/* the code object of conv3x3_wino4b_f32 (build/wino4b_gfx950.hsaco) as a byte blob of libmiunet.so; assembled from build/ */
	.section .rodata
	.balign 4096
	.globl miunet_wino4b_hsaco
	.globl miunet_wino4b_hsaco_end
miunet_wino4b_hsaco:
	.incbin "wino4b_gfx950.hsaco"
miunet_wino4b_hsaco_end:
	.section .note.GNU-stack,"",@progbits
