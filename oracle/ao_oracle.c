/* placeholder so the Makefile target exists; AO oracle added with row a1 */
int orc_ao_placeholder(void) { return 0; }
