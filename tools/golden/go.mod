module groot-golden

go 1.14

require (
	github.com/ekzhu/lshensemble v1.1.0
	github.com/will-rowe/groot v1.1.2
	github.com/will-rowe/nthash v0.2.0
)
